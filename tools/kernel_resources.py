#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    python -m nclone_amd.build_native -v 2> build.log ; python tools/kernel_resources.py build.log [out.csv]

(The numbers are those of the code objects that ship in nclone_amd/libnpp_amd.so: same compiler invocation.)"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return [re.sub(r"npp::\(anonymous namespace\)::", "", o) for o in out[:len(names)]]
    except Exception:
        return names


def parse(txt):
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append([b.split("\n")[0].strip(), g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                     g(r"Occupancy \[waves/SIMD\]"), g("SGPRs Spill"), g("VGPRs Spill"), g(r"LDS Size \[bytes/block\]")])
    names = demangle([r[0] for r in rows])
    for r, n in zip(rows, names):
        r[0] = n
    return rows


if __name__ == "__main__":
    rows = parse(open(sys.argv[1]).read())
    hdr = ["kernel", "vgpr", "agpr", "sgpr", "scratch_bytes_per_lane", "occupancy_waves_per_simd", "sgpr_spill", "vgpr_spill", "static_lds"]
    lines = [",".join(hdr)] + [",".join('"%s"' % v if i == 0 else str(v) for i, v in enumerate(r)) for r in rows]
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write("\n".join(lines) + "\n")
    for r in rows:
        print("%-66s vgpr %3d agpr %3d sgpr %3d scratch %4d occ %d sspill %3d vspill %3d" % tuple([r[0][:66]] + r[1:8]))
