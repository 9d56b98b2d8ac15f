#!/bin/bash
export TMPDIR=/tmp
python tools/probes/stage_b2b.py 9
