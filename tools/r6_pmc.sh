# r6 (as r5): per-launch counters of the three roofline kernels at bench.py's shapes (GPU box) -> gpurun_out/r6_pmc.json (copied to
# profiles/r6_pmc.json: bench.py reads it).  HBM / fabric traffic = FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes
# (MI355X_MICROARCH.md's HBM section: KB units, FETCH x2 on gfx950); for the ray-marcher additionally the issue counters that say
# what it is bound by (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_VALU_MFMA_BUSY_CYCLES, SQ_WAVE_CYCLES, GRBM_GUI_ACTIVE, ...).  Every
# entry carries the sha256 of the kernel source it was measured on.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6_pmc
rm -rf $OUT; mkdir -p $OUT
run() { n=$1; c=$2; shift 2; timeout 180 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$n.$(echo $c | tr ' ' '+') -- python $R/tools/pmc_one.py "$@" > /dev/null 2>&1; }
for c in FETCH_SIZE WRITE_SIZE; do
  run gemm_fc1_gelu_12288x4096x1024 $c gemm 12288 4096 1024
  run gemm_fc1_gelu_49152x4096x1024 $c gemm 49152 4096 1024
  run gemm_fc2_gateres_12288x1024x4096 $c gemm_gr 12288 1024 4096
  run attention_256x768x768x64 $c attn 256 768 768 64
  run attention_1024x768x1024x64 $c attn 1024 768 1024 64
  run render_4x256 $c render 4 256
done
run render_4x256 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" render 4 256
run render_4x256 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" render 4 256
run render_4x256 "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" render 4 256
python3 - <<PY
import csv, glob, collections, hashlib, json, os
R, OUT = "$R", "$OUT"
want = {"gemm": ("gemm_bf16_", "ln3diff_amd/csrc/gemm_bf16.hip"), "attention": ("attn_", "ln3diff_amd/csrc/attention.hip"),
        "render": ("render_kernel", "ln3diff_amd/csrc/render.hip")}
res = {}
for d in sorted(glob.glob(OUT + "/*")):
    key, ctrs = os.path.basename(d).rsplit(".", 1)
    pat, hip = want[key.split("_")[0]]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not vals:
        print("NO DATA", d); continue
    name, cs = max(vals.items(), key=lambda kv: sum(sum(v) for v in kv[1].values()))
    e = res.setdefault(key, {"kernel": name[:120], "hip": hip, "sha16": hashlib.sha256(open(os.path.join(R, hip), "rb").read()).hexdigest()[:16], "counters": {}})
    for c, v in cs.items():
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            e[c + "_kb_mean"] = sum(v) / len(v)
        else:
            e["counters"][c] = sum(v) / len(v)
        e["launches"] = len(v)
for k, e in res.items():
    f, w = e.get("FETCH_SIZE_kb_mean"), e.get("WRITE_SIZE_kb_mean")
    if f is not None and w is not None:
        # MI355X_MICROARCH.md: both counters are in KB; FETCH_SIZE under-reports by 2x on gfx950 (64 B requests counted as 32 B)
        e["traffic_bytes"] = int(f * 1024 * 2 + w * 1024)
        e["source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB, FETCH x2 gfx950 correction), tools/r6_pmc.sh"
        print("%-36s fetch %.1f MB (x2 corrected) write %.1f MB  -> %.1f MB / launch" % (k, f * 2 / 1024, w / 1024, e["traffic_bytes"] / 1e6))
    c = e["counters"]
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_ACTIVE_INST_* count quad-cycles per wave (x4 = cycles); 1024 SIMDs
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        e["issue"] = {"cycles_per_launch": cyc, "valu_insts": c["SQ_INSTS_VALU"],
                      "valu_busy_frac": c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (cyc * 1024),
                      "mfma_busy_frac": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024),
                      "lds_busy_frac": c.get("SQ_ACTIVE_INST_LDS", 0) * 4 / (cyc * 1024),
                      "valu_issue_floor_frac": c["SQ_INSTS_VALU"] * 4 / (cyc * 1024)}
        print(k, "issue:", json.dumps(e["issue"]))
json.dump(res, open(R + "/gpurun_out/r6_pmc.json", "w"), indent=1)
PY
