#!/bin/bash
# On the GPU box: stand-alone time of the six CFConv launches of a step (tools/nodeconv_time.py) for one molecule x 25 .. 280
# conformers with 4, 2 and 1 targets per wave -- where the batch-size thresholds of BatchTopology.group_targets come from.
cd $GRAFT_REPO_ROOT
for cp in 25 70 100 140 200 280; do for g in 4 2 1; do
python3 tools/nodeconv_time.py --mols 1 --copies $cp --group $g --only node 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('copies $cp N',d['N'],'group',d['group_targets'],'tiles',d['local_tiles'],'node_x6_ms %.4f'%d['node_x6_ms'])"
done; done
