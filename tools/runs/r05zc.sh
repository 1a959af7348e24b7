R=$GRAFT_REPO_ROOT
grep -m1 "model name" /proc/cpuinfo; grep -m1 flags /proc/cpuinfo | tr ' ' '\n' | grep -c -E "^(bmi2|adx|avx2|avx512ifma)$"
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_fixed_base.py tests/test_gpu_msm_small.py -q -m gpu -x 2>&1 | tail -2
for ch in 8 4; do echo "--- ATLAS_MSM_CHUNK=$ch"; ATLAS_MSM_CHUNK=$ch LOG_N=22 TAB_C=0 timeout 300 python tools/time_msm_tab.py | grep "table c"; done
echo "--- 2^24, 2^20"; LOG_N=24 TAB_C=0 timeout 300 python tools/time_msm_tab.py | grep "table c"; LOG_N=20 TAB_C=0 timeout 300 python tools/time_msm_tab.py | grep "table c"
cd /tmp && export TMPDIR=/tmp
for ch in 8 4; do
ATLAS_MSM_CHUNK=$ch LOG_N=22 TAB_C=0 timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_msm$ch -o msm -- python $R/tools/time_msm_tab.py > /dev/null 2>&1
DB=$(find /tmp/prof_msm$ch -name "*.db" | head -1)
python $R/tools/rocprof_timeline.py $DB 8 > $R/gpurun_out/r05zc_msm_timeline_$ch.txt 2>&1
cat $R/gpurun_out/r05zc_msm_timeline_$ch.txt
done
