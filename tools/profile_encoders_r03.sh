#!/bin/bash
# round 3: SQ counters (MFMA busy, LDS conflicts) of the precision mode that meets the tolerance (f16x3): the CNN encoder's 32x32 kernels
# (fused last layer), the generic convolution (U-Net) and the training step's kernels -> gpurun_out/prof_enc_r03/pmc_summary.json
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_enc_r03
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
C1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
C2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_UNALIGNED_STALL"
i=0
for C in "$C1" "$C2"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace -d $OUT -o cnn_p$i --output-format csv -- python $R/tools/run_hip_encoder_f16x3.py 1024 hip_f16x3 > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc $C --kernel-trace -d $OUT -o unet_p$i --output-format csv -- python $R/tools/run_unet.py 1024 f16x3 1 > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc $C --kernel-trace -d $OUT -o train_p$i --output-format csv -- python $R/tools/train_step_profile.py 1024 f16x3 1 > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, json
res = {}
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "nastar_conv" in k or "wgrad_kernel" in k or "chan_" in k:
            k = k.split("(")[0][-80:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    tag = "cnn" if "cnn_p" in f else ("unet" if "unet_p" in f else "train")
    for k, d in acc.items():
        res.setdefault(tag + " | " + k, {}).update({c: v for c, v in d.items()})
        res[tag + " | " + k]["launches"] = max(cnt[k].values())
for k, d in res.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"] > 0:
        # SQ_BUSY_CYCLES sums over the 32 shader engines (8 XCD x 4), SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (the r01 / r02 definition)
        d["mfma_busy_frac_of_simd_cycles"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * d["SQ_BUSY_CYCLES"])
    if "SQ_LDS_BANK_CONFLICT" in d and d.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        d["lds_conflict_frac"] = d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]
json.dump(res, open("$OUT/pmc_summary.json", "w"), indent=1)
for k, d in res.items():
    print(k, {c: (round(v, 4) if isinstance(v, float) and v < 10 else int(v)) for c, v in d.items() if c in ("mfma_busy_frac_of_simd_cycles", "lds_conflict_frac", "launches", "SQ_INSTS_MFMA", "SQ_WAIT_INST_LDS")})
PY
rm -f $OUT/*kernel_trace.csv
