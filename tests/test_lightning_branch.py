"""The Lightning-driven path (reference: osu_dreamer/scripts/fit_denoiser.py:23-32 hands `DiffusionTrainer` to a
`pytorch_lightning.Trainer`).  Lightning is not installed on the MI355X image, so `DiffusionTrainer` normally derives from
`nn.Module`; this test puts a minimal stand-in `pytorch_lightning` on the import path of a CHILD process (the package must not have been
imported yet) and checks the branch that is otherwise never executed: the class derives from `LightningModule`, the constructor
records its hyper-parameters through `save_hyperparameters`, `training_step` / `validation_step` log through `self.log_dict` once a
trainer is attached, and the hooks run in Lightning's order for two steps on the emulator build."""
import os
import subprocess
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = '''
import torch
class LightningModule(torch.nn.Module):
    """the slice of pytorch_lightning.LightningModule that DiffusionTrainer touches"""
    def __init__(self):
        super().__init__()
        self.saved_hparams, self.logged = None, []
        self._trainer = None
    def save_hyperparameters(self, *a, **k):
        import inspect
        frame = inspect.currentframe().f_back
        self.saved_hparams = {k: v for k, v in frame.f_locals.items() if k not in ("self", "__class__")}
    def log_dict(self, d, *a, **k):
        self.logged.append(dict(d))
class LightningDataModule:
    pass
'''

CHILD = '''
import os, sys
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests")); sys.path.insert(0, {stubdir!r})
import pytorch_lightning as pl
import torch
from kernel_backend import EMU_SO, build_emu
build_emu()
from osu_dreamer_amd import _lib, train
_lib.use_library(EMU_SO)
assert train.HAVE_LIGHTNING and issubclass(train.DiffusionTrainer, pl.LightningModule)
from oracle import denoiser_oracle as O
from osu_dreamer_amd.lr_schedule import LRScheduleArgs
from osu_dreamer_amd.model import BackboneArgs, DiffusionModelArgs
d = O.TINY
tr = train.DiffusionTrainer(val_batches=2, opt_args=dict(lr=3e-4, weight_decay=0.01),
                            schedule_args=LRScheduleArgs(warmup_init=.3, warmup_steps=10, decay_start=100), osl_weight=1., del_weight=30.,
                            emb_dim=d.emb_dim, a_dim=d.a_dim, style_dim=d.style_dim,
                            diffusion_args=DiffusionModelArgs(d.global_cond_dim, d.backbone_dim,
                                                              BackboneArgs(d.depth, d.expand, d.head_dim, d.n_heads, d.radius), d.u_head_dim))
assert set(tr.saved_hparams) >= {{"val_batches", "opt_args", "schedule_args", "osl_weight", "del_weight", "emb_dim", "a_dim", "style_dim", "diffusion_args"}}
tr.diffusion.load_state_dict(O.init_params(d, seed=5))
tr._trainer = object()                         # "attached to a Trainer": logging goes through log_dict
data = O.synthetic_batch(d, 2, 24, seed=6)
batch = (data["h"], data["z"], data["s"], torch.zeros(2, 5))
cfg = tr.configure_optimizers()
opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
assert cfg["lr_scheduler"]["interval"] == "step"
losses = []
for i in range(2):                             # Lightning's per-batch order
    opt.zero_grad()
    loss = tr.training_step(batch, i)
    loss.backward()
    opt.step(); sched.step()
    tr.on_train_batch_end()
    losses.append(float(loss.detach()))
assert all(l == l and l > 0 for l in losses) and int(tr.diffusion_ema.n_averaged) == 2
assert len(tr.logged) == 2 and set(tr.logged[0]) == {{"train/loss", "train/osl", "train/del", "train/u_mape"}}
full = (data["h"][:1], data["z"][:1], data["s"][:1], torch.zeros(1, 5))
tr.validation_step(full, 0)
assert set(tr.logged[-1]) == {{"val/loss", "val/osl", "val/del", "val/u_mape"}}
print("LIGHTNING_BRANCH_OK", losses)
'''


def test_lightning_branch_runs(tmp_path):
    stubdir = tmp_path / "stubs"
    (stubdir / "pytorch_lightning").mkdir(parents=True)
    (stubdir / "pytorch_lightning" / "__init__.py").write_text(STUB)
    script = tmp_path / "child.py"
    script.write_text(textwrap.dedent(CHILD.format(repo=REPO, stubdir=str(stubdir))))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "LIGHTNING_BRANCH_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
