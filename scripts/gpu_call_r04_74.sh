#!/bin/bash
cd "$GRAFT_REPO_ROOT"
TRACE_ONE_PASS=1 TRACE_WINDOW_MS=35.5 bash scripts/trace_size.sh 16384 1024 r04_onepass_N16384
