"""Experiment: where does k_conv_fused_unit spend its time?  (FSW_FUSED_ABL bits, see conv_fused.hip)"""
import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for dbg in (0, 1, 2, 4, 6, 7):
    env = dict(os.environ, FSW_FUSED_ABL=str(dbg))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True).stdout
    import json
    try:
        d = json.loads(out.strip().splitlines()[-1])
    except Exception:
        print("dbg=%d: no result" % dbg); continue
    print("dbg=%d fused=%.3f ms embed_only=%.3f ms" % (dbg, d["stage_ms"]["conv_fused_unit"], d["stage_ms"]["embed_reg_unit"]), flush=True)
