"""Library GEMM selections for the encoder shapes (PyTorch TunableOp results, gfx950).

The encoders' forward and dX products stay on hipBLASLt / rocBLAS.  For several of their shapes the library's
default heuristic does not pick its own fastest kernel; ``gemm_gfx950.csv`` holds the winners TunableOp measured on an MI355X
for the [batch x tokens, 768 / 2304 / 3072] products of ViT-B/16 and BERT-base at per-GPU batch 1024 and 256 and (round 5,
``tools/tune_gemms.py --workload ijepa_vitl``) for the ViT-L/16 and 384-wide predictor products of the I-JEPA step at batch 128, every
mask geometry that occurs (4 x 128 x (169 + 30 .. 40) predictor rows), and for HTSAT's plain products at batch 256
(``tools/probes/record_untuned_three_tower.py`` lists them without tuning -- no strided-batched op is left in that tower since its
windowed attention is one kernel -- and ``tools/tune_gemms.py --untuned`` tunes exactly that list offline).  ``enable()`` loads
them (no tuning at run time, nothing written); a shape that is not in the file runs the library's default, and a file recorded
with another PyTorch / hipBLASLt / rocBLAS build is refused by TunableOp's validators, which leaves every shape on the default.

Recording more shapes::

    PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=/tmp/t.csv python your_step.py

and merge the ``GemmTunableOp`` / ``GemmAndBiasTunableOp`` lines.  (Strided-batched entries are left out on purpose: tuning
the 24 x 64 x 64 batched products of the HTSAT tower's window attention ends in a GPU memory fault inside one of the library's
candidate kernels, HISTORY.md 5.)
"""

import os
import warnings

import torch

DEFAULT_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_gfx950.csv")


def enable(path: str = None) -> bool:
    """Switch TunableOp to look-up-only mode and load ``path`` (default: the shipped file, or ``$MMK_TUNED_FILE``).  Returns
    whether the selections were accepted."""
    import torch.cuda.tunable as tunable

    if path is None:
        path = os.environ.get("MMK_TUNED_FILE", DEFAULT_FILE)

    if not torch.cuda.is_available():
        raise RuntimeError("mmlearn_amd.tuned.enable() needs the GPU (the selections name gfx950 library kernels)")
    tunable.enable(True)
    tunable.tuning_enable(False)
    ok = bool(tunable.read_file(path))
    if not ok:
        tunable.enable(False)
        # not silent: without the selections the encoder GEMMs run the library's default heuristic (about 3 ms of a 200 ms step)
        warnings.warn(f"mmlearn_amd.tuned: {path} was refused by TunableOp's validators (recorded with another PyTorch / hipBLASLt / "
                      "rocBLAS build?); library GEMMs stay on the default heuristic", RuntimeWarning, stacklevel=2)
    return ok


def disable() -> None:
    import torch.cuda.tunable as tunable

    tunable.enable(False)
