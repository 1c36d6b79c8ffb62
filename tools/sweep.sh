#!/bin/bash
python tools/time_parts.py --what mstep --tag "mstep accumulate" 2>/dev/null | tail -1
python tools/time_parts.py --what mstep_out --tag "mstep + finalize" 2>/dev/null | tail -1
python tools/time_parts.py --what estep --tag "estep" 2>/dev/null | tail -1
python tools/time_parts.py --what step --tag "step (M + epoch_end)" 2>/dev/null | tail -1
