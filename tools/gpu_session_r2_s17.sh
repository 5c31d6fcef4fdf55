#!/bin/bash
# Round 2, session 17: re-check of the two test files touched after r2s16 (restructured COCODetection, oracle cache in the
# batch-parity tests): pull_item tests + the R50 batch-8 parity cases + cross-process determinism.
O=gpurun_out/r2s17; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_jpeg.py "tests/test_gpu_batch_parity.py::test_timed_plan_matches_oracle_at_batch[configs1_r50_b8]" tests/test_gpu_batch_parity.py::test_forced_f4x4_plan_matches_oracle_at_batch8 tests/test_gpu_batch_parity.py::test_plan_is_deterministic_across_processes -m gpu -q --timeout 500 -rA --durations=8 > $O/pytest.log 2>&1; grep -E "passed|failed|s call" $O/pytest.log | tail -12; grep -E "^FAILED|^ERROR" $O/pytest.log | head
