mkdir -p gpurun_out/r03h; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in A B; do
  if [ $v = A ]; then export SEPFINDER_LIB=$PWD/multi_robot_slam_separators_amd/libsepfinder_A.so; else unset SEPFINDER_LIB; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03h/cnn_$v -- python3 tools/netvlad_latency.py 640 480 20 > gpurun_out/r03h/cnn_$v.log 2>&1
done
python3 - <<'PY'
import csv,glob,re
for v in "AB":
    f=glob.glob("gpurun_out/r03h/cnn_%s/*/*kernel_trace.csv"%v)[0]
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    idx=[i for i,r in enumerate(rows) if "k_conv3x3_first" in r["Kernel_Name"]]
    seq=rows[idx[-1]:]
    out=[]
    for r in seq:
        n=r["Kernel_Name"]; m=re.search(r"(k_[a-z0-9_]+)",n); nm=m.group(1) if m else n[:20]
        t=re.search(r"ILi(\d+)ELi(\d+)E",n)
        d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
        out.append("%s%s %.0f" % (nm.replace("k_conv_igemm_h","h").replace("k_",""), ("%sx%s"%t.groups()) if t else "", d))
    print(v, " | ".join(out[:24]))
PY
