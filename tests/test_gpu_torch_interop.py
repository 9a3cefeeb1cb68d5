"""Zero-copy hand-over of a result buffer to torch (what bench.py's RCCL gather relies on)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import sys, numpy as np
    sys.path.insert(0, %r)
    import torch                                   # torch first, as under torchrun in bench.py
    from pyimsegm_amd import _hip
    rng = np.random.default_rng(0)
    labels = rng.integers(0, 40, (120, 200)).astype(np.int32)
    lut = rng.integers(0, 3, 40).astype(np.int32)
    sess = _hip.Image2D(120, 200).set_labels(labels)
    segm, _ = sess.gather(lut, None)
    t = torch.as_tensor(_hip.segm_device_array(sess), device='cuda')
    assert t.dtype == torch.int32 and tuple(t.shape) == (120, 200)
    assert np.array_equal(t.cpu().numpy(), lut[labels]) and np.array_equal(segm, lut[labels])
    print('INTEROP_OK')
''') % ROOT


def test_device_array_to_torch():
    res = subprocess.run([sys.executable, '-c', SCRIPT], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and 'INTEROP_OK' in res.stdout, res.stdout[-1500:] + res.stderr[-1500:]
