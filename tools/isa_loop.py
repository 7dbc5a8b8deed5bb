#!/usr/bin/env python3
"""tools/isa_loop.py -- where do a kernel's instructions sit?  Reads a hipcc -S device listing, takes one kernel
(substring of its mangled name) and prints, per basic block between labels, the instruction mix (VALU, SALU, LDS,
global, scratch, lane moves, waits), flagging blocks that sit inside a loop (a later branch jumps back to or above
them).  Used to check that spills stay out of the hot loop and to count a tile's instructions.

    hipcc --offload-arch=gfx950 ... --cuda-device-only -S -o k.s file.hip ; tools/isa_loop.py k.s k_fir_fusedILi0ELi2ELi1
"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or (key in l and re.match(r"^_Z\S+:", l)))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    blocks, cur, name = [], [], "entry"
    label_at = {}
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((name, cur))
            name, cur = m.group(1), []
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur.append(t)
    blocks.append((name, cur))
    for i, (n, _) in enumerate(blocks):
        label_at[n] = i
    # loops: a branch in block i to label j <= i marks blocks j..i as in a loop (depth counts nesting)
    depth = [0] * len(blocks)
    for i, (n, ins) in enumerate(blocks):
        for t in ins:
            m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t)
            if m:
                tgt = m.group(1) or m.group(2)
                j = label_at.get(tgt)
                if j is not None and j <= i:
                    for k in range(j, i + 1):
                        depth[k] += 1
    tot = {}
    print(f"{'block':>12} {'depth':>5} {'n':>5} {'valu':>5} {'salu':>5} {'lds':>4} {'glob':>4} {'scr':>4} {'lane':>4} {'wait':>4} {'f64':>4}")
    for i, (n, ins) in enumerate(blocks):
        c = dict(n=len(ins), valu=0, salu=0, lds=0, glob=0, scr=0, lane=0, wait=0, f64=0)
        for t in ins:
            op = t.split()[0]
            if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
                c["lane"] += 1
            elif op.startswith("v_"):
                c["valu"] += 1
                if "_f64" in op:
                    c["f64"] += 1
            elif op.startswith("s_waitcnt"):
                c["wait"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"):
                c["glob"] += 1
            elif op.startswith("scratch_"):
                c["scr"] += 1
        if c["n"] >= int(sys.argv[3]) if len(sys.argv) > 3 else c["n"] >= 12:
            print(f"{n:>12} {depth[i]:>5} {c['n']:>5} {c['valu']:>5} {c['salu']:>5} {c['lds']:>4} {c['glob']:>4} {c['scr']:>4} {c['lane']:>4} {c['wait']:>4} {c['f64']:>4}")
        for k, v in c.items():
            tot.setdefault(depth[i], {}).setdefault(k, 0)
            tot[depth[i]][k] += v
    for d in sorted(tot):
        print("depth", d, tot[d])


if __name__ == "__main__":
    main()
