#!/bin/bash
timeout -k 10 400 python tools/convt_slots_check.py 64 2 2>&1 | grep -v amdgpu.ids | cut -c1-250
