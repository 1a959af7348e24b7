python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r05final_bench.json 2> gpurun_out/r05final_bench.err; tail -c 300 gpurun_out/r05final_bench.json
