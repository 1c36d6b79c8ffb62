#!/bin/bash
python tools/time_parts.py --what thr --tag "threshold+truncate N=65536" 2>&1 | tail -1
python tools/time_parts.py --what thr --n 8192 --tag "threshold+truncate N=8192" 2>&1 | tail -1
