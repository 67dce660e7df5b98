#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into small files under
profiles/: kernel stats CSV rows for our kernels, HBM traffic per launch (FETCH_SIZE / WRITE_SIZE with the
gfx950 corrections of MI355X_MICROARCH.md "HBM"), SQ counter means."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def counter_means(path, kernel="npp_step_kernel", last=None):
    """Mean counter value per dispatch of `kernel`; `last` = only the last N dispatches (the bench's timed region: the launches
    before it are the pre-roll, during which npp_step's autotuner rotates through the kernel's build variants)."""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    names = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            agg[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
            names[int(r["Dispatch_Id"])].add(r["Kernel_Name"])
    out = {}
    for c, d in agg.items():
        ids = sorted(d)
        if last:
            ids = ids[-last:]
        out[c] = {"dispatches": len(ids), "mean_per_dispatch": sum(d[i] for i in ids) / len(ids),
                  "kernels": sorted({k for i in ids for k in names[i]})}
    return out


TIMED = 300   # tools/profile_round.sh runs bench.py --steps 300


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    out = {"tag": tag}
    ks = find(os.path.join(src, "trace"), "*kernel_stats.csv")
    if ks:
        rows = [r for r in csv.DictReader(open(ks)) if "npp_" in r["Name"]]
        with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
        for r in rows:
            for key, pat in (("step_kernel", "npp_step_kernel"), ("render_kernel", "npp_render_kernel")):
                if pat in r["Name"]:
                    out[key] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                "max_ns": float(r["MaxNs"])}
    kt = find(os.path.join(src, "trace"), "*kernel_trace.csv")
    if kt and "step_kernel" in out:
        # the bench's timed region = the LAST `steps` dispatches of the step kernel (warm-up launches come first)
        rows = [r for r in csv.DictReader(open(kt)) if "npp_step_kernel" in r["Kernel_Name"]]
        steps = 300
        try:
            steps = json.loads(open(os.path.join(src, "bench_line.json")).read())["steps"]
        except Exception:
            pass
        last = rows[-steps:]
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last]
        out["step_kernel"]["timed_region_avg_ns"] = sum(d) / len(d)
        out["step_kernel"]["timed_region_launches"] = len(d)
        out["step_kernel"]["timed_region_kernel"] = last[-1]["Kernel_Name"]
        # launch descriptor as the kernel trace reports it (arch VGPRs and AGPRs are separate columns; LDS is the dynamic +
        # static allocation of the dispatch; scratch in bytes per lane)
        for col, name in (("Grid_Size_X", "grid_size_x"), ("Workgroup_Size_X", "workgroup_size_x"), ("VGPR_Count", "arch_vgpr"),
                          ("Accum_VGPR_Count", "agpr"), ("SGPR_Count", "sgpr"), ("Scratch_Size", "scratch_bytes_per_lane"),
                          ("LDS_Block_Size", "lds_bytes_per_workgroup")):
            out["step_kernel"][name] = last[0].get(col)
        rr = [r for r in csv.DictReader(open(kt)) if "npp_render_kernel" in r["Kernel_Name"]]
        if rr and "render_kernel" in out:
            d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rr[-steps:]]
            out["render_kernel"]["timed_region_avg_ns"] = sum(d) / len(d)
            for col, name in (("Grid_Size_X", "grid_size_x"), ("VGPR_Count", "arch_vgpr"), ("Accum_VGPR_Count", "agpr"),
                              ("Scratch_Size", "scratch_bytes_per_lane"), ("LDS_Block_Size", "lds_bytes_per_workgroup")):
                out["render_kernel"][name] = rr[-1].get(col)
    traffic = {}
    for name in ("fetch", "write"):
        cc = find(os.path.join(src, name), "*counter_collection.csv")
        if cc:
            traffic.update(counter_means(cc, last=TIMED))
    if traffic:
        # FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B (hbm_bytes = value * 1024); on gfx950
        # FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams (x2 correction).  This kernel's reads
        # are 8-byte-per-lane plane reads (uncalibrated width), so both the raw and the x2 figure are recorded.
        fs = traffic.get("FETCH_SIZE", {}).get("mean_per_dispatch")
        ws = traffic.get("WRITE_SIZE", {}).get("mean_per_dispatch")
        out["traffic"] = {
            "FETCH_SIZE_mean": fs, "WRITE_SIZE_mean": ws,
            "read_bytes_per_launch_raw": None if fs is None else fs * 1024,
            "read_bytes_per_launch_x2": None if fs is None else fs * 2048,
            "write_bytes_per_launch": None if ws is None else ws * 1024,
        }
        if fs is not None and ws is not None:
            out["traffic"]["hbm_bytes_per_launch"] = fs * 2048 + ws * 1024
    rtraffic = {}
    for name in ("fetch", "write"):
        cc = find(os.path.join(src, name), "*counter_collection.csv")
        if cc:
            rtraffic.update(counter_means(cc, "npp_render_kernel", last=TIMED))
    if rtraffic:
        # the render kernel reads with dword loads and writes with dword stores: FETCH_SIZE recorded raw and x2 (the x2
        # correction is calibrated for 16 B / lane streams only); WRITE_SIZE is exact for coalesced stores
        fs = rtraffic.get("FETCH_SIZE", {}).get("mean_per_dispatch")
        ws = rtraffic.get("WRITE_SIZE", {}).get("mean_per_dispatch")
        out["render_traffic"] = {"FETCH_SIZE_mean": fs, "WRITE_SIZE_mean": ws,
                                 "read_bytes_per_launch_raw": None if fs is None else fs * 1024,
                                 "read_bytes_per_launch_x2": None if fs is None else fs * 2048,
                                 "write_bytes_per_launch": None if ws is None else ws * 1024}
        if fs is not None and ws is not None:
            out["render_traffic"]["hbm_bytes_per_launch"] = fs * 2048 + ws * 1024
    # every kernel of ours in the run (round 3: the config-5 profile holds five): timed-region duration from the kernel trace and HBM
    # traffic from the two PMC passes, per launch
    by_kernel = {}
    if kt:
        rows_all = list(csv.DictReader(open(kt)))
        for key, pat in (("step", "npp_step_kernel"), ("player_frame", "npp_render_kernel"), ("global_view", "npp_global_view_kernel"),
                         ("reachability", "npp_reach_kernel"), ("switch_states", "npp_switch_states_kernel")):
            rr = [r for r in rows_all if pat in r["Kernel_Name"]]
            if not rr:
                continue
            d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rr[-TIMED:]]
            e = {"launches": len(d), "timed_region_avg_ns": sum(d) / len(d), "kernel": rr[-1]["Kernel_Name"][:120],
                 "arch_vgpr": rr[-1].get("VGPR_Count"), "agpr": rr[-1].get("Accum_VGPR_Count"),
                 "scratch_bytes_per_lane": rr[-1].get("Scratch_Size"), "lds_bytes_per_workgroup": rr[-1].get("LDS_Block_Size")}
            t = {}
            for name in ("fetch", "write"):
                cc = find(os.path.join(src, name), "*counter_collection.csv")
                if cc:
                    t.update(counter_means(cc, pat, last=TIMED))
            fs = t.get("FETCH_SIZE", {}).get("mean_per_dispatch")
            ws = t.get("WRITE_SIZE", {}).get("mean_per_dispatch")
            if fs is not None and ws is not None:
                e.update({"read_bytes_per_launch_raw": fs * 1024, "read_bytes_per_launch_x2": fs * 2048, "write_bytes_per_launch": ws * 1024,
                          "hbm_bytes_per_launch": fs * 2048 + ws * 1024})
            by_kernel[key] = e
        # the step kernel's build variants (npp_set_step_variant) apart: the pre-roll's autotuner windows run all three, and the variant of
        # the timed region can differ between the trace pass and a PMC pass -- bench.py picks the row of the variant IT ran
        import re
        variants = {}
        for name in ("fetch", "write"):
            cc = find(os.path.join(src, name), "*counter_collection.csv")
            if not cc:
                continue
            acc = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in csv.DictReader(open(cc)):
                mt = re.search(r"npp_step_kernel<16, (true|false), false, false, (\d)>", r["Kernel_Name"])
                if mt and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    acc[mt.group(2)][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
            for v, d in acc.items():
                ids = sorted(d)[-TIMED:]
                variants.setdefault(v, {})["FETCH_SIZE" if name == "fetch" else "WRITE_SIZE"] = (sum(d[i] for i in ids) / len(ids), len(ids))
        sv = {}
        for v, t in variants.items():
            if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
                fs, ws = t["FETCH_SIZE"][0], t["WRITE_SIZE"][0]
                sv[v] = {"dispatches": min(t["FETCH_SIZE"][1], t["WRITE_SIZE"][1]), "read_bytes_per_launch_raw": fs * 1024,
                         "read_bytes_per_launch_x2": fs * 2048, "write_bytes_per_launch": ws * 1024, "hbm_bytes_per_launch": fs * 2048 + ws * 1024}
        if sv and "step" in by_kernel:
            by_kernel["step"]["variants"] = sv
            mt = re.search(r"false, false, (\d)>", by_kernel["step"]["kernel"])
            if mt and mt.group(1) in sv:   # the row of the variant the TRACE pass timed (a PMC pass may have settled on another)
                by_kernel["step"].update({k: v for k, v in sv[mt.group(1)].items() if k != "dispatches"})
                by_kernel["step"]["traffic_of_variant"] = int(mt.group(1))
    if by_kernel:
        out["by_kernel"] = by_kernel
    sq = find(os.path.join(src, "sq"), "*counter_collection.csv")
    if sq:
        out["sq"] = counter_means(sq, last=TIMED)
        rs = counter_means(sq, "npp_render_kernel", last=TIMED)
        if rs:
            out["sq_render"] = rs
    bl = os.path.join(src, "bench_line.json")
    if os.path.isfile(bl) and os.path.getsize(bl):
        out["bench_line_under_profiler"] = json.loads(open(bl).read())
    with open(os.path.join(dst, "%s_summary.json" % tag), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main()
