cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_s2
timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "sp8 or sp4" 2>&1 | tail -120 > gpurun_out/r06_s2/pytest_8rank.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "sp2-10 or ti2v" 2>&1 | tail -120 > gpurun_out/r06_s2/pytest_sp2.log
tail -5 gpurun_out/r06_s2/*.log
