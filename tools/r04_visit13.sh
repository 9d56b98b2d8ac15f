#!/bin/bash
export TMPDIR=/tmp
bash tools/ab_env.sh "PA_CHAIN_WPW=4" "PA_CHAIN_WPW=6" 2>&1 | grep -v "^sa0.fps"
