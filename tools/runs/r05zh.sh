R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
LOG_T=16 ATLAS_NO_PIPELINE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ra -o ra -- python $R/tools/trace_ra_small.py 2>&1 | grep "us/round"
DB=$(find /tmp/prof_ra -name "*.db" | head -1)
python $R/tools/rocprof_timeline.py $DB 130 > $R/gpurun_out/r05zh_ra_timeline.txt 2>&1
LOG_T=16 timeout 300 python $R/tools/trace_ra_small.py 2>&1 | grep "us/round" | tail -2
