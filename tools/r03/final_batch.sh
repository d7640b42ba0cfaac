#!/bin/bash
# end-of-round check on the GPU box: smoke, the whole GPU suite, the default bench line
cd "$(dirname "$0")/../.."
O=gpurun_out/r03
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
( time python -m pytest tests -m gpu -q -x ) 2>&1 | tail -12 > $O/gputest.txt; cat $O/gputest.txt
