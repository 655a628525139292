"""The value + gradient of the log-density of the cfg4 CNN alone (for rocprofv3 --kernel-trace --stats): GRADS gradient
calls after one warm-up, nothing else on the device apart from the set-up."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ["CFG4_SETUP_ONLY"] = "1"
import cfg4_cnn_bench as m  # noqa: E402

z = np.zeros(m.M)
m.ctx.logdensity_grad(z)
for _ in range(int(os.environ.get("GRADS", "5"))):
    m.ctx.logdensity_grad(z)
