"""The code paths that exist only when ``mmlearn`` is importable, executed against the REAL reference classes.

``hydra_zen`` / ``lightning`` are absent from the image, so in every other test ``mmlearn_amd.tasks.base`` takes its stand-alone
branch and ``registry.BACKEND`` is ``"none"``.  Here a fresh interpreter installs the import shim the golden generator uses
(``tests/golden/ref_shim.py``: stand-ins for hydra_zen / lightning / torchmetrics / timm, the reference itself untouched) BEFORE
importing this package, so that

* ``mmlearn_amd.tasks.TrainingTask`` derives from ``mmlearn.tasks.base.TrainingTask`` and ``configure_optimizers`` IS the reference's
  (mmlearn/tasks/base.py:72-155), run on this package's ``ContrastivePretraining``;
* ``EvaluationHooks`` is the reference's; the registry registers through a store object instead of recording only.

Lightning itself is still a stand-in (the real ``Trainer`` has never driven these classes: INTEGRATION.md).  Needs ``/root/reference``:
skipped where it does not exist (the GPU box); nothing in the product, the GPU tests, ``smoke()`` or ``bench.py`` reads it."""

import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import ref_shim  # noqa: E402

SCRIPT = textwrap.dedent('''
    import sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, {golden!r})
    import ref_shim
    ref_shim.install()
    from functools import partial
    import torch
    import mmlearn.tasks.base as ref_base
    import mmlearn.tasks.hooks as ref_hooks
    import mmlearn_amd, mmlearn_amd.tasks as T
    from mmlearn_amd.tasks import base, ContrastivePretraining, IJEPA, LossPairSpec
    from mmlearn_amd import registry, ContrastiveLoss

    assert base.HAVE_MMLEARN is True
    assert issubclass(T.TrainingTask, ref_base.TrainingTask) and issubclass(ContrastivePretraining, ref_base.TrainingTask)
    assert issubclass(IJEPA, ref_base.TrainingTask) and issubclass(T.EvaluationHooks, ref_hooks.EvaluationHooks)
    assert T.TrainingTask.configure_optimizers is ref_base.TrainingTask.configure_optimizers      # inherited, not restated
    assert registry.BACKEND != "none", registry.BACKEND
    assert ("modules/losses", "ContrastiveLossHIP") in registry.REGISTERED and ("task", "ContrastivePretrainingHIP") in registry.REGISTERED

    class Enc(torch.nn.Module):
        def __init__(self, key, d_in):
            super().__init__()
            self.key, self.net = key, torch.nn.Sequential(torch.nn.Flatten(1), torch.nn.Linear(d_in, 8), torch.nn.LayerNorm(8))
        def forward(self, inputs):
            return (self.net(inputs[self.key]),)

    task = ContrastivePretraining(encoders={{"rgb": Enc("rgb", 12), "text": Enc("text", 5)}}, loss=ContrastiveLoss(),
                                  optimizer=partial(torch.optim.AdamW, lr=1e-3, weight_decay=0.2),
                                  lr_scheduler={{"scheduler": partial(torch.optim.lr_scheduler.StepLR, step_size=1), "extras": {{"interval": "step"}}}},
                                  modality_loss_pairs=[LossPairSpec(("rgb", "text"))], compute_validation_loss=False, compute_test_loss=False)
    out = task.configure_optimizers()            # the reference's own implementation
    groups = {{g["name"]: g for g in out["optimizer"].param_groups}}
    assert len(groups["weight_decay_params"]["params"]) == 2 and groups["weight_decay_params"]["weight_decay"] == 0.2
    assert len(groups["no_weight_decay_params"]["params"]) == 2 * 3 + 1 and groups["no_weight_decay_params"]["weight_decay"] == 0.0   # biases, LN, log_logit_scale
    assert out["lr_scheduler"]["interval"] == "step"
    # the reference's constructor contract (mmlearn/tasks/base.py:60-64) reaches through this package's subclass
    try:
        ContrastivePretraining(encoders={{"rgb": Enc("rgb", 12)}}, loss=None)
        raise SystemExit("expected ValueError")
    except ValueError:
        pass
    # the encoders run on the host through the task's own encode() (the HIP ops need a GPU: only the kernel-free part is exercised here)
    e = task.encode({{"rgb": torch.randn(4, 12)}}, mmlearn_amd.modalities.Modalities.get_modality("rgb"), normalize=False)
    assert e.shape == (4, 8)
    task.log("x", torch.tensor(1.0))             # the (stand-in) LightningModule's log, reached through the reference base class
    print("BINDING-OK", registry.BACKEND)
''')


@pytest.mark.skipif(not ref_shim.reference_available(), reason="needs the read-only reference at /root/reference (build container only)")
def test_tasks_derive_from_the_reference_base_classes_when_mmlearn_is_importable():
    code = SCRIPT.format(root=ROOT, golden=os.path.join(ROOT, "tests", "golden"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "BINDING-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
