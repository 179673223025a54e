"""Style-model sampler on the HIP path vs the reference's own outputs (tests/golden/style_*.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import style_oracle as SO
from osu_dreamer_amd.style import StyleModel, StyleModelArgs
from kernel_backend import dev, rel_l2  # noqa: F401

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["style_tiny", "style_full"])
def test_style_forward_and_sampler(dev, name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    fx = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
    v_ = [int(x) for x in fx["dims"].tolist()]
    d = SO.StyleDims(style_dim=v_[0], label_features=v_[1], h_dim=v_[2], depth=v_[3], expand=v_[4])
    if dev.type == "cpu" and d.h_dim > 64:
        pytest.skip("full-width style model runs on the GPU only")
    P = ({k[2:]: t for k, t in fx.items() if k.startswith("w.")} if any(k.startswith("w.") for k in fx)
         else SO.init_style_params(d, int(fx["seed"])))
    m = StyleModel(d.style_dim, StyleModelArgs(d.label_features, d.h_dim, d.depth, d.expand))
    assert sorted(m.state_dict().keys()) == sorted(P.keys())
    m.load_state_dict(P)
    m = m.to(dev)
    labels, st = fx["labels"].to(dev), fx["st"].to(dev)
    assert rel_l2(m.compute_conditioning(labels), fx["cond"]) < 1e-5
    u, v = m(st, labels)
    assert rel_l2(u, fx["fwd_u"]) < 1e-5 and rel_l2(v, fx["fwd_v"]) < 2e-5
    s = m.sample(labels, 16, s_init=fx["s_init"].to(dev))
    assert rel_l2(s, fx["sample_s"]) < 1e-4
    if dev.type == "cuda":
        m.use_graph = False
        assert torch.equal(m.sample(labels, 16, s_init=fx["s_init"].to(dev)), s)
