import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from svpc_amd import synthetic
orig = synthetic.make_config
ph = float(os.environ.get("EXP_PH", "0.1")); pa = float(os.environ.get("EXP_PA", "0.1"))
def mk(**kw):
    kw.setdefault("hidden_dropout_prob", ph); kw.setdefault("attention_probs_dropout_prob", pa)
    return orig(**kw)
synthetic.make_config = mk
import svpc_amd
svpc_amd.make_config = mk
bench.main()
