#!/bin/bash
# round 4, call 39: smoke() on the final sources
cd /root/repo
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 5
