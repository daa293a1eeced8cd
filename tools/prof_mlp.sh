cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/mlp_bench.py 20 2>&1 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mlp_trace -- python3 tools/mlp_bench.py 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/mlp_pmc1 -- python3 tools/mlp_bench.py 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/mlp_pmc2 -- python3 tools/mlp_bench.py 3 > /dev/null 2>&1
find gpurun_out/mlp_trace -name "*kernel_stats.csv" | head -1 | xargs head -5
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/mlp_pmc1", "gpurun_out/mlp_pmc2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "psfnet_mlp" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(d.split("/")[-1], k, "n", len(v), "mean", sum(v) / len(v))
PY
