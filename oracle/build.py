"""
Build recipe for the C oracle (TEST INFRASTRUCTURE -- see oracle/__init__.py).

    python -m oracle.build          # -> oracle/_build/liboracle.so

-ffp-contract=off is mandatory: the float paths must round once per reference op.
There is no oracle/_ref here: the reference is pure Python on TensorFlow, there is no
C/C++ source under /root/reference to compile.
"""

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, '_build')
LIB = os.path.join(OUT_DIR, 'liboracle.so')
SRC = os.path.join(HERE, 'oracle.c')


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    cmd = ['gcc', '-O2', '-std=c11', '-fPIC', '-shared', '-fopenmp', '-ffp-contract=off',
           '-fno-fast-math', '-Wall', '-Wextra', SRC, '-o', LIB, '-lm']
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    print(build(force=True))
