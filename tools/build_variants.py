import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g, os, sys
from concurrent.futures import ThreadPoolExecutor
V = {"wide64": ("HILC_RES_WIDE_MASK=1",), "wide96": ("HILC_RES_WIDE_MASK=2",), "wide128": ("HILC_RES_WIDE_MASK=4",)}
if len(sys.argv) > 1:      # name=DEF[,DEF...] ...
    V = {a.split("=", 1)[0]: tuple(a.split("=", 1)[1].split(",")) for a in sys.argv[1:]}
def one(kv):
    name, defs = kv
    try:
        g.compile_library(os.path.join(g.LIBDIR, f"libv_{name}.so"), defines=defs, only=("resblock.hip", "resblock_chain.hip", "rvq.hip"))
        return name, "ok"
    except Exception as e:
        return name, "FAILED " + str(e)[:100]
with ThreadPoolExecutor(4) as ex:
    for r in ex.map(one, V.items()):
        print(r)
