#!/bin/bash
# tools/collect_profiles.sh ROUND  (GPU box, from the repo root) -- the evidence bench.py's figures rest on, into
# gpurun_out/profiles_ROUND/ (tools/publish_profiles.sh copies what is to be judged into profiles/):
#   bench.json                      the plain bench line (unprofiled)
#   c2box/ c2fir/ c3/ c4/ c4fir/ c1/  rocprofv3 --kernel-trace --stats of the bench command for each workload, then the
#                                   PMC passes (separate runs: counters are never combined with tracing)
#   summary_*.txt                   tools/pmc_summary.py over each
#   traffic.json                    HBM bytes per launch of the dominant kernels (2 x FETCH_SIZE + WRITE_SIZE, gfx950 note
#                                   in MI355X_MICROARCH.md), keyed as bench.py looks them up, with the hash of the sources
set -u
R=${1:-r03}
ROOT=$PWD
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p "$OUT"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
prof() { # name, bench args...
  local name=$1; shift
  tools/prof.sh "$OUT/$name" -- python3 "$ROOT/bench.py" --no-extra --no-cpu-baseline "$@" > /dev/null 2>&1
  python3 tools/pmc_summary.py "$OUT/$name" > "$OUT/summary_$name.txt" 2>&1
  echo "profiled $name"
}
prof c2box
prof c2fir --resampler fir
prof c3 --workload C3
prof c4 --workload C4 --frames 16
prof c4fir --workload C4 --frames 32 --resampler fir
prof c1 --workload C1 --frames 256
python3 - "$OUT" <<'PY'
import hashlib, json, os, re, sys
out = sys.argv[1]
sha = hashlib.sha256(open("hdr2yuv_amd/libhdr2yuv_hip.so", "rb").read()).hexdigest()
import importlib.util
spec = importlib.util.spec_from_file_location("bench_mod", "bench.py")
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
res = {"_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/prof.sh) on `python3 bench.py --no-extra --no-cpu-baseline ...`; "
               "per-dispatch average of the dominant kernel; FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 reports half the bytes of wide coalesced "
               "16-B/lane reads), both in KiB", "library_sha256": sha, "sources_sha256": bench.sources_sha256()}
F = 128  # bench.py's default frames per step (C4 box 16, C4 FIR 32, C1 256: given on the command lines above, as bench.py's `others` run them)
alg = {"c2box": (f"C2_box_F{F}", 15.0 * 3840 * 2160 * F), "c2fir": (f"C2_fir_F{F}", 15.0 * 3840 * 2160 * F), "c3": (f"C3_box_F{F}", 18.0 * 3840 * 2160 * F),
       "c4": ("C4_box_F16", 9.0 * 7680 * 4320 * 16), "c4fir": ("C4_fir_F32", 9.0 * 7680 * 4320 * 32), "c1": ("C1_box_F256", 15.0 * 1920 * 1080 * 256)}
for name, (key, ab) in alg.items():
    try:
        txt = open(os.path.join(out, f"summary_{name}.txt")).read()
    except OSError:
        continue
    best = None
    for blk in txt.split("== ")[1:]:
        head = blk.splitlines()[0]
        if not head.startswith("void k_f"):
            continue
        f = re.search(r"FETCH_SIZE\s+total\s+\d+\s+per-dispatch\s+([\d.]+)\s+\((\d+) dispatches", blk)
        w = re.search(r"WRITE_SIZE\s+total\s+\d+\s+per-dispatch\s+([\d.]+)", blk)
        if f and w and (best is None or int(f.group(2)) > best[3]):
            best = (head, float(f.group(1)), float(w.group(1)), int(f.group(2)))
    if best:
        hb = (2 * best[1] + best[2]) * 1024
        res[key] = {"kernel": best[0], "FETCH_SIZE_KiB": best[1], "WRITE_SIZE_KiB": best[2], "hbm_bytes_per_launch": int(hb),
                    "algorithmic_bytes_per_launch": int(ab), "ratio": round(hb / ab, 3), "dispatches": best[3], "library_sha256": sha}
json.dump(res, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
# pictures unlike the headline's; transfer pairs at 8 frames per launch; the stream pipeline
bash tools/contentbench.sh "$OUT/content" > "$OUT/content.txt" 2>&1
python3 tools/tfbench.py > "$OUT/tfbench.txt" 2>&1
bash tools/densebench.sh "$OUT/dense" > "$OUT/dense.txt" 2>&1
python3 tools/streambench.py > "$OUT/streambench.txt" 2>&1
