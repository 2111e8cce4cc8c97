#!/bin/bash
# full GPU suite + smoke, then the default bench line
mkdir -p gpurun_out
bash scripts/r4/run19.sh || exit 1
bash scripts/r4/run15.sh
