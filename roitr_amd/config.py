"""Config loading, mirroring configs/utils.py:4-18 of the reference: one YAML file whose sections are
flattened into a single dict (later keys win), wrapped so that keys read as attributes (main.py:46 EasyDict)."""
import os

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))


class Config(dict):
    """dict with attribute access (stand-in for easydict.EasyDict, which is not a dependency here)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def load_config(path):
    """One flat dict out of the YAML's sections (a key that appears in two sections takes the later one)."""
    with open(path, "r") as f:
        sections = yaml.safe_load(f)
    return {key: value for section in sections.values() for key, value in section.items()}


def test_config(benchmark="3DMatch"):
    """The test-mode settings of configs/test/tdmatch.yaml / fdmatch.yaml (values restated in roitr_amd/configs)."""
    name = "tdmatch_test.yaml" if benchmark in ("3DMatch", "3DLoMatch") else "fdmatch_test.yaml"
    c = Config(load_config(os.path.join(_HERE, "configs", name)))
    c.benchmark = benchmark
    return c
