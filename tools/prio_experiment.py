"""Does the priority of the HIP queues change how the two streams of the step share the chip?  Runs bench.py's timed loop with the whole step on a
stream of priority NEMAR_MAIN_PRIORITY (default: torch's current stream) and the side stream at NEMAR_SIDE_PRIORITY.
usage: NEMAR_MAIN_PRIORITY=-1 NEMAR_SIDE_PRIORITY=0 python tools/prio_experiment.py [bench args]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
print('priority range', torch.cuda.Stream.priority_range(), file=sys.stderr)
import bench
mp = os.environ.get('NEMAR_MAIN_PRIORITY')
if mp is None:
    bench.main()
else:
    s = torch.cuda.Stream(priority=int(mp))
    with torch.cuda.stream(s):
        bench.main()
