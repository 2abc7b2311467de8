"""CPU oracle for the mmlearn contrastive / I-JEPA hot path.

TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import anything under ``oracle/``; the
product package ``mmlearn_amd`` never does and fails loudly when its HIP
library is missing.
"""
