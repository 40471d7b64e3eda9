#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q -k "altcorr" 2>&1 | tail -8
