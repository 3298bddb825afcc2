import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g, os, sys
from concurrent.futures import ThreadPoolExecutor
V = {"kp8d2": ("HILC_RES_KP=8", "HILC_RES_DEPTH=2"), "kp4d2": ("HILC_RES_KP=4", "HILC_RES_DEPTH=2"), "kp8d3": ("HILC_RES_KP=8", "HILC_RES_DEPTH=3"),
     "rb2": ("HILC_RES_RB=2",)}
def one(kv):
    name, defs = kv
    try:
        g.compile_library(os.path.join(g.LIBDIR, f"libv_{name}.so"), defines=defs, only=("resblock.hip", "resblock_chain.hip"))
        return name, "ok"
    except Exception as e:
        return name, "FAILED " + str(e)[:100]
with ThreadPoolExecutor(4) as ex:
    for r in ex.map(one, V.items()):
        print(r)
