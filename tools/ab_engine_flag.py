#!/usr/bin/env python
"""A/B of a module-level switch of hilcodec_amd.engine through bench.py, same process setup as the product:
   python tools/ab_engine_flag.py STREAM_STAGE0=0 --mode streaming --graph --no-cpu-baseline --no-other-configs"""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import hilcodec_amd.engine as e
name, val = sys.argv[1].split("=")
assert hasattr(e, name), name
setattr(e, name, type(getattr(e, name))(int(val)))
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
