#!/usr/bin/env python3
"""Rewrites the switch table of INTEGRATION.md section 6 from the libraries' own tables (gec_env_table / gbm_env_table)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(lib, fn):
    f = getattr(ctypes.CDLL(os.path.join(ROOT, "garage_amd", lib)), fn)
    f.restype = ctypes.c_char_p
    out = []
    for line in f().decode().splitlines():
        if line.strip():
            name, default, meaning = line.split("\t", 2)
            out.append(f"| `{name}` | {default} | {meaning} |")
    return out


def main():
    ctypes.CDLL(os.path.join(ROOT, "garage_amd", "libgarage_ec.so"), mode=ctypes.RTLD_GLOBAL)
    table = ["| switch | default | meaning |", "|---|---|---|"] + rows("libgarage_ec.so", "gec_env_table") + rows("libgarage_block.so", "gbm_env_table")
    p = os.path.join(ROOT, "INTEGRATION.md")
    s = open(p).read()
    m = re.search(r"\| switch \| default \| meaning \|\n(\|.*\n)+", s)
    s = s[:m.start()] + "\n".join(table) + "\n" + s[m.end():]
    open(p, "w").write(s)
    print(len(table) - 2, "switches")


if __name__ == "__main__":
    main()
