"""SURVEY.md s8 row f1 (replay sampling + n-step targets) inside the driver's GPU record: the host-side parity tests of
tests/test_replay_native.py need no GPU, so the `-m "not gpu"` run is where they normally execute; these wrappers run the
same checks under the gpu marker, so that GPUTEST_rNN re-confirms them on the GPU box's host (where the replay actually
runs beside the device loop) -- plus the parallel ingest against one thread there."""
import os

import pytest

from tests import test_replay_native as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('path', T.FILES, ids=[os.path.basename(f)[:-4] for f in T.FILES])
def test_sample_batch_matches_reference_on_the_gpu_host(path):
  T.test_sample_batch_matches_reference(path)


@pytest.mark.parametrize('path', T.FILES, ids=[os.path.basename(f)[:-4] for f in T.FILES])
def test_save_history_matches_reference_on_the_gpu_host(path):
  T.test_save_history_matches_reference(path)


def test_parallel_ingest_is_bit_identical_on_the_gpu_host():
  T.test_parallel_ingest_is_bit_identical_to_one_thread(40, 500, 3000)
