cd /root/repo; mkdir -p gpurun_out/r5ab
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5ab/pytest.txt 2>&1; tail -n 4 gpurun_out/r5ab/pytest.txt
timeout 900 python tools/stress_super.py --reps 20 > gpurun_out/r5ab/stress.txt 2>&1; tail -n 2 gpurun_out/r5ab/stress.txt
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -1
