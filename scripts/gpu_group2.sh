#!/bin/bash
cd /root/repo
SGA_ADJ_STATS=1 timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | tail -14 | head -13 | cut -c1-140
