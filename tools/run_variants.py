#!/usr/bin/env python3
"""usage: tools/run_variants.py <config letter(s) of run_configs.py> -- run the config with every library under variants/"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in sorted(os.listdir(os.path.join(root, "variants"))):
    env = dict(os.environ, MMD_LIB_DIR=os.path.join(root, "variants", name))
    for rep in range(2):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "run_configs.py")] + sys.argv[1:], env=env, capture_output=True, text=True)
        print("%-20s %s" % (name, (r.stdout.strip() or r.stderr.strip()[-300:])), flush=True)
