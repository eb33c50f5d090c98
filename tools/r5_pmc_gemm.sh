# r5: re-collect the GEMM entries of profiles/r5_pmc.json after csrc/gemm_bf16.hip changed (ring configuration 13 added): same passes as
# tools/r5_pmc.sh for the three GEMM keys, merged into the existing record (attention / render entries keep their own sha).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5_pmc_gemm
rm -rf $OUT; mkdir -p $OUT
run() { n=$1; c=$2; shift 2; timeout 180 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$n.$c -- python $R/tools/pmc_one.py "$@" > /dev/null 2>&1; }
for c in FETCH_SIZE WRITE_SIZE; do
  run gemm_fc1_gelu_12288x4096x1024 $c gemm 12288 4096 1024
  run gemm_fc1_gelu_49152x4096x1024 $c gemm 49152 4096 1024
  run gemm_fc2_gateres_12288x1024x4096 $c gemm_gr 12288 1024 4096
done
python3 - <<PY
import csv, glob, collections, hashlib, json, os
R, OUT = "$R", "$OUT"
hip = "ln3diff_amd/csrc/gemm_bf16.hip"
rec = json.load(open(os.path.join(R, "profiles/r5_pmc.json")))
new = {}
for d in sorted(glob.glob(OUT + "/*")):
    key, ctr = os.path.basename(d).rsplit(".", 1)
    vals = collections.defaultdict(list)
    name = None
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_bf16_ring64_kernel" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    if not vals:
        print("NO DATA", d); continue
    name, v = max(vals.items(), key=lambda kv: sum(kv[1]))
    e = new.setdefault(key, {"kernel": name[:120], "hip": hip, "sha16": hashlib.sha256(open(os.path.join(R, hip), "rb").read()).hexdigest()[:16], "counters": {}})
    e[ctr + "_kb_mean"] = sum(v) / len(v); e["launches"] = len(v)
for k, e in new.items():
    f, w = e.get("FETCH_SIZE_kb_mean"), e.get("WRITE_SIZE_kb_mean")
    if f is None or w is None:
        print("INCOMPLETE", k); continue
    e["traffic_bytes"] = int(f * 1024 * 2 + w * 1024)
    e["source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB, FETCH x2 gfx950 correction), tools/r5_pmc_gemm.sh"
    print("%-36s fetch %.1f MB (x2 corrected) write %.1f MB  -> %.1f MB / launch (was %.1f)" % (k, f * 2 / 1024, w / 1024, e["traffic_bytes"] / 1e6, rec.get(k, {}).get("traffic_bytes", 0) / 1e6))
    rec[k] = e
json.dump(rec, open(os.path.join(R, "gpurun_out/r5_pmc.json"), "w"), indent=1)
PY
