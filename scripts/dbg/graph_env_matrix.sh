#!/bin/bash
# Does a HIP-runtime graph setting change the graph-vs-eager mismatch of the pad64 build? (diagnosis only)
export MADIFF_PAD64=1
run() { echo "== $*"; env "$@" timeout -k 10 120 python -u scripts/dbg/graph_vs_eager.py 1 2>&1 | grep -E "block  7|^loss" ; }
run A=1 || exit 1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 || exit 1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1 || exit 1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1 || exit 1
