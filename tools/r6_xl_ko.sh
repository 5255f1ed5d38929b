#!/bin/bash
# timing-only knock-outs of k_sparse_large<XL>'s row loops (tools/probe_xl_timeline.py, GNNX_PROBE_KO): which part of a round bounds it?
cd $GRAFT_REPO_ROOT
for ko in "" fwdstores gestores bwdloads; do
  echo "== knock-out: ${ko:-none}"
  GNNX_PROBE_KO=$ko timeout 300 python tools/probe_xl_timeline.py 2146 2>&1 | grep -E "node|layer 1 on|dX1|per near|iteration"
done
