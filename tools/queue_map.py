#!/usr/bin/env python3
"""Which HSA queue did every kernel of a rocprofv3 --kernel-trace run go to?  python3 tools/queue_map.py <kernel_trace.csv>
Prints, per queue id, the launch count by kernel name (short), and the number of distinct queues the composite kernels used."""
import csv, re, sys
from collections import Counter, defaultdict
q = defaultdict(Counter)
for r in csv.DictReader(open(sys.argv[1])):
    name = re.match(r"(?:void )?([A-Za-z_0-9:]+)", r["Kernel_Name"]).group(1)
    q[r["Queue_Id"]][name] += 1
for k, c in sorted(q.items()):
    print("queue", k, dict(c.most_common(6)))
print("composite_bwd launches by queue:", {k: c["composite_bwd_kernel"] for k, c in q.items() if c["composite_bwd_kernel"]})
