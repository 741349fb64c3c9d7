"""Derived kernel tables are what their generators say (CPU, no device)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_i4_selector_table_is_generated_from_the_prediction_table():
    # k_i4_sel (the pool gather of wave_i4_choose) against k_i4_lut (H.264 8.3.1.2 written out): same table, same predictions
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_i4_sel.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "agree" in r.stdout
