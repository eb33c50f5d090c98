# HBM / fabric traffic per launch of the three roofline kernels at bench.py's shapes (GPU box): FETCH_SIZE and WRITE_SIZE in
# SEPARATE rocprofv3 --pmc passes (MI355X_MICROARCH.md's HBM section), written as gpurun_out/r4_pmc.json with the sha256 of the
# kernel source each number belongs to.  Copy that file to profiles/r4_pmc.json: bench.py reads it (pmc_traffic()).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4_pmc
rm -rf $OUT; mkdir -p $OUT
run() { n=$1; c=$2; shift 2; timeout 180 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$n.$c -- python $R/tools/pmc_one.py "$@" > /dev/null 2>&1; }
for c in FETCH_SIZE WRITE_SIZE; do
  run gemm_fc1_gelu_12288x4096x1024 $c gemm 12288 4096 1024
  run gemm_fc1_gelu_49152x4096x1024 $c gemm 49152 4096 1024
  run attention_256x768x768x64 $c attn 256 768 768 64
  run attention_1024x768x1024x64 $c attn 1024 768 1024 64
  run render_4x256 $c render 4 256
done
python3 - <<PY
import csv, glob, collections, hashlib, json, os
R, OUT = "$R", "$OUT"
want = {"gemm": ("gemm_bf16_ring64_kernel", "ln3diff_amd/csrc/gemm_bf16.hip"), "attention": ("attn_", "ln3diff_amd/csrc/attention.hip"),
        "render": ("render_kernel", "ln3diff_amd/csrc/render.hip")}
res = {}
for d in sorted(glob.glob(OUT + "/*")):
    key, ctr = os.path.basename(d).rsplit(".", 1)
    pat, hip = want[key.split("_")[0]]
    vals = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    if not vals:
        print("NO DATA", d); continue
    name, v = max(vals.items(), key=lambda kv: sum(kv[1]))
    e = res.setdefault(key, {"kernel": name[:120], "hip": hip, "sha16": hashlib.sha256(open(os.path.join(R, hip), "rb").read()).hexdigest()[:16]})
    e[ctr + "_kb_mean"] = sum(v) / len(v)
    e["launches"] = len(v)
for k, e in res.items():
    f, w = e.get("FETCH_SIZE_kb_mean"), e.get("WRITE_SIZE_kb_mean")
    if f is None or w is None: continue
    # MI355X_MICROARCH.md: both counters are in KB; FETCH_SIZE under-reports by 2x on gfx950 (64 B requests counted as 32 B)
    e["traffic_bytes"] = int(f * 1024 * 2 + w * 1024)
    e["source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB, FETCH x2 gfx950 correction), tools/pmc_traffic.sh"
    print("%-36s fetch %.1f MB (x2 corrected) write %.1f MB  -> %.1f MB / launch" % (k, f * 2 / 1024, w / 1024, e["traffic_bytes"] / 1e6))
json.dump(res, open(R + "/gpurun_out/r4_pmc.json", "w"), indent=1)
PY
