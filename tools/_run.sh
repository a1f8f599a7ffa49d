#!/bin/bash
# scratch entry point for one-off gpurun experiments (kept so that `gpurun -- 'bash tools/_run.sh'` always exists)
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
