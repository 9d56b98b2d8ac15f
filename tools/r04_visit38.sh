#!/bin/bash
cd /root/repo
echo "== default"; python tools/f16_cos.py 2>&1 | grep -v amdgpu
echo "== PA_ENGINE_VLAD_F16=0"; PA_ENGINE_VLAD_F16=0 python tools/f16_cos.py 2>&1 | grep -v amdgpu
echo "== PA_ENGINE_VLAD_F16=0 PA_ENGINE_FPX16=0"; PA_ENGINE_VLAD_F16=0 PA_ENGINE_FPX16=0 python tools/f16_cos.py 2>&1 | grep -v amdgpu
