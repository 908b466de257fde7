for v in ${VARIANTS}; do
export TMPDIR=/tmp; rm -rf gpurun_out/prof_sorv
PCP_HIP_LIB=$PWD/point-cloud-processing_amd/lib/libpcp_hip_$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sorv -- python3 tools/bench_configs.py --frames 64 > /dev/null 2>&1
echo == $v
python3 - <<P
import csv,glob
f=glob.glob("gpurun_out/prof_sorv/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "sor_knn" in r["Name"]: print(r["Name"][:40], r["Calls"], round(float(r["AverageNs"])/1e3,1))
P
done
