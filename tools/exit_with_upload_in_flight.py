#!/usr/bin/env python3
"""A process that ends while an upload is still in flight: starts the native uploader on 2 GB of float64 and leaves at once.
    python tools/exit_with_upload_in_flight.py [daemon]     (daemon: the finish thread as it was until round 6's last day)
The exit code is the test: 0 = a clean exit (the interpreter waited for the upload), -6 / 134 = SIGABRT at exit."""
import os
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from litcoder_core_amd import ops  # noqa: E402

if "daemon" in sys.argv[1:]:
    _init = threading.Thread.__init__

    def init(self, *a, **k):
        if k.get("name") == "lc-upload-finish":
            k["daemon"] = True
        _init(self, *a, **k)
    threading.Thread.__init__ = init
dev = ops.device(0)
T, V = 3000, 80000
Y = np.random.default_rng(0).standard_normal((T, V))
dY = torch.empty((T, ops.pad_to(V, 128)), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
up = ops.PanelUploader([(ops.HostRows([Y]), dY, a, b) for a, b in ((0, 20000), (20000, 40000), (40000, 60000), (60000, V))], dev)
up.wait(0)
print("first panel issued; leaving", flush=True)
sys.exit(0)
