bash tools/capture_extra.sh r05zz > /dev/null 2>&1
python tools/gpt2_by_operator.py gpt2 2 > gpurun_out/r05zz_gpt2_by_operator.txt 2>&1
head -22 gpurun_out/r05zz_gpt2_by_operator.txt
bash tools/runs/full_suite.sh r05zz
