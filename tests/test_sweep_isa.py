"""The IC(0) sweep kernels issue their record loads by hand (inline asm) and retire them with
hand-counted s_waitcnt; tools/check_sweep_isa.py proves on the generated gfx950 ISA, over every
control-flow path, that no instruction touches an operand register while its load is in flight."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_in_flight_operand_is_touched():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_sweep_isa.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" 0 touches of an in-flight operand") == 4, r.stdout
