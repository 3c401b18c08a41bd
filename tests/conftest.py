import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")
WEIGHTS = os.path.join(REPO, "weights")

# golden fixture name -> (weights stem, future, iso)
VARIANTS = {
    "basic-iso3200": ("recurrent-convunet-iso3200", 0, 3200),
    "basic-future-iso3200": ("recurrent-convunet-future-iso3200", 1, 3200),
    "feat-iso3200": ("recurrent-convunet+feat-iso3200", 0, 3200),
    "feat-future-iso12800": ("recurrent-convunet+feat-future-iso12800", 1, 12800),
    "next-iso3200": ("recurrent-ConvNeXtUnet-iso3200", 0, 3200),
    "next-feat-future-iso3200": ("recurrent-ConvNeXtUnet+feat-future-iso3200", 1, 3200),
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_weights(stem):
    from safetensors.torch import load_file
    return load_file(os.path.join(WEIGHTS, stem + ".safetensors"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
