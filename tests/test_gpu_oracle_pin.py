"""The oracle's pin, re-run in the GPU tier.

tests/test_oracle_golden.py (CPU, `-m "not gpu"`) checks the CPU restatement against the reference-generated fixtures of
tests/golden/.  Every GPU parity test compares the HIP path with that oracle, so the `-m gpu` run on the MI355X box would
otherwise never execute the pin it rests on.  This file runs the same functions again under the gpu marker (they need no GPU
and take a few seconds): a GPU-tier record is then self-contained -- reference fixtures -> oracle -> HIP path -- on one box.

Cross-host mode: the fixtures were generated in the authoring container; on the GPU box's host CPU torch's vectorised kernels
round a few table entries one ulp differently (measured: 5 of the 22 cases), so float comparisons run with
test_oracle_golden.CROSS_HOST = True (rtol 2e-4); token ids, index tables, masks and hashes stay exact.
"""
import itertools

import pytest

import test_oracle_golden as pin

pytestmark = pytest.mark.gpu


def _cases():
    out = []
    for name in sorted(n for n in dir(pin) if n.startswith("test_")):
        fn = getattr(pin, name)
        marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
        if not marks:
            out.append(pytest.param(fn, {}, id=name))
            continue
        axes = []
        for m in marks:
            names = [a.strip() for a in m.args[0].split(",")] if isinstance(m.args[0], str) else list(m.args[0])
            axes.append([dict(zip(names, v if len(names) > 1 else (v,))) for v in m.args[1]])
        for combo in itertools.product(*axes):
            kw = {}
            for d in combo:
                kw.update(d)
            out.append(pytest.param(fn, kw, id=name + "[" + "-".join(str(v) for v in kw.values()) + "]"))
    return out


CASES = _cases()


def test_the_pin_is_complete():
    assert len(CASES) >= 20, len(CASES)


@pytest.mark.parametrize("fn,kwargs", CASES)
def test_oracle_pin(fn, kwargs, monkeypatch):
    monkeypatch.setattr(pin, "CROSS_HOST", True)
    fn(**kwargs)
