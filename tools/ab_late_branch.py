import runpy, sys
import hilcodec_amd.engine as e
e.STREAM_LATE_BRANCH = False
sys.argv = ["bench.py", "--mode", "streaming", "--graph", "--no-cpu-baseline", "--no-other-configs", "--no-clock-probe"]
runpy.run_path("bench.py", run_name="__main__")
