# SQ counters of the NetVLAD trunk's convolution kernels, last inference only (no autotune trials): run on the GPU box
mkdir -p gpurun_out/cnn_pmc; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/cnn_pmc/run -- python3 tools/netvlad_latency.py 640 480 3 > gpurun_out/cnn_pmc/run.log 2>&1
python3 - <<'PY'
import csv,glob,re
from collections import defaultdict, OrderedDict
f=glob.glob("gpurun_out/cnn_pmc/run/*/*counter_collection.csv")[0]
rows=list(csv.DictReader(open(f)))
disp=OrderedDict()
for r in rows:
    d=disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": r["Grid_Size"]})
    d[r["Counter_Name"]]=float(r["Counter_Value"])
ids=sorted(disp)
first=[i for i in ids if "k_conv3x3_first" in disp[i]["name"]]
for i in [j for j in ids if j>=first[-1]]:
    d=disp[i]; n=d["name"]
    if "igemm_h" not in n: continue
    m=re.search(r"ILi(\d+)ELi(\d+)E",n)
    cyc=d["GRBM_GUI_ACTIVE"]/8
    print("h%sx%s grid %8s  cyc/XCD %7.0f  mfma_busy %.2f  valu_busy %.2f  lds_inst/mfma %.2f  parked %.2f  issue-stalled %.2f" % (m.group(1),m.group(2),d["grid"],cyc,d["SQ_VALU_MFMA_BUSY_CYCLES"]/(1024*cyc),d["SQ_INSTS_VALU"]*4/(1024*cyc),d["SQ_INSTS_LDS"]/max(1,d["SQ_INSTS_MFMA"]),d["SQ_WAIT_ANY"]/d["SQ_WAVE_CYCLES"],d["SQ_WAIT_INST_ANY"]/d["SQ_WAVE_CYCLES"]))
PY
