#!/usr/bin/env python3
"""Loop-level instruction mix of a kernel in hipcc's -save-temps assembly (.s): for every backward branch, the static
instructions between its target label and the branch -- spills (scratch_*), SGPR spill traffic (v_readlane / v_writelane),
AGPR copies, s_nop, fp64 ops.  Used to check that the depenetration loop of npp_step_kernel holds no spill code.

    python tools/isa_loops.py file.s <kernel-name-substring>"""
import re
import sys


def main():
    path, want = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    # kernel body: from "<mangled>:" to ".Lfunc_end"
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w*:", l) and want in l:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {}
    insts = []
    for l in body:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        insts.append(t.split(";")[0].strip())
    print("kernel %s: %d instructions" % (lines[start].strip(), len(insts)))
    loops = []
    for i, t in enumerate(insts):
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)|^s_branch\s+(\.LBB\d+_\d+)", t)
        if m:
            lab = m.group(1) or m.group(2)
            if lab in labels and labels[lab] <= i:
                loops.append((labels[lab], i, lab))
    def mix(a, b):
        seg = insts[a:b + 1]
        c = lambda pat: sum(1 for t in seg if re.match(pat, t))
        return {"n": len(seg), "scratch": c(r"scratch_"), "lane": c(r"v_(read|write)lane"), "acc": c(r"v_accvgpr"), "nop": c(r"s_nop"),
                "f64": c(r"v_\w+_f64"), "rsq": c(r"v_rsq_f64"), "rcp": c(r"v_rcp_f64"), "bperm": c(r"ds_bpermute"), "dpp": sum(1 for t in seg if "dpp" in t),
                "saveexec": c(r"s_(and|or)_saveexec"), "div": c(r"v_div_(scale|fmas|fixup)")}
    for a, b, lab in sorted(loops, key=lambda x: x[1] - x[0]):
        m = mix(a, b)
        if m["n"] < 40:
            continue
        print("loop %-10s [%5d..%5d] %s" % (lab, a, b, " ".join("%s=%d" % kv for kv in m.items())))
    m = mix(0, len(insts) - 1)
    print("whole kernel:", " ".join("%s=%d" % kv for kv in m.items()))


if __name__ == "__main__":
    main()
