#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in "1" "0"; do echo "== build=$a"; timeout 300 python tools/dbg_half2.py $a 2>&1 | grep -v "^  File\|^$" | tail -12; done
