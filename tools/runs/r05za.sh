R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_tab.py -q -m gpu -x 2>&1 | tail -3
echo "--- fold: old (scalar multiples)"; ATLAS_MSM_FOLD_MUL=1 LOG_N=22 TAB_C=0 timeout 300 python tools/time_msm_tab.py | grep table
for lo in 12 11 10 9 8; do echo "--- ATLAS_TAB_LO=$lo"; ATLAS_TAB_LO=$lo LOG_N=22 TAB_C=0 timeout 300 python tools/time_msm_tab.py | grep "table c"; done
cd /tmp && export TMPDIR=/tmp
LOG_N=22 TAB_C=0 timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_msm -o msm -- python $R/tools/time_msm_tab.py > /dev/null 2>&1
DB=$(find /tmp/prof_msm -name "*.db" | head -1)
python $R/tools/rocprof_timeline.py $DB 24 > $R/gpurun_out/r05za_msm_timeline.txt 2>&1
cat $R/gpurun_out/r05za_msm_timeline.txt
