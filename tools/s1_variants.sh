export TMPDIR=/tmp
for v in $VARIANTS; do
rm -rf gpurun_out/prof_scv
PCP_HIP_LIB=$PWD/point-cloud-processing_amd/lib/libpcp_hip_$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_scv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-mode --no-verify > /dev/null 2>&1
echo == $v
python3 - <<P
import csv,glob
f=glob.glob("gpurun_out/prof_scv/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k1_" in r["Name"]: print(r["Name"][:30], r["Calls"], round(float(r["AverageNs"])/1e3,1))
P
done
