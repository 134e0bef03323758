"""Token-id parity of the AR decode, where the device and the oracle disagree (test helper, not a test file).

north_star asks for bit-exact LLM token ids.  Bit-exactness needs bit-identical logits; the device's bf16 GEMV reduction order
differs from the CPU oracle's, so on a flat (random-weight) next-token distribution a few draws land on the other side of a
CDF boundary.  This helper turns "a few flips are tolerated" into "every flip is accounted for":

torch.multinomial(p, 1) is argmax_i p_i / q_i with q ~ Exp(1) (ATen; tests/test_gpu_stages.py::
test_fused_sampling_equals_torch_multinomial pins the device sampler to it draw for draw).  The oracle is teacher-forced on the
device's own history and draws q from the same generator state, so step s of both sides sees the same prefix and the same q.
If the device picked a and the oracle b != a, then
    log p_dev[a] - log p_dev[b]  >=  log q[a] - log q[b]  >=  log p_ref[a] - log p_ref[b],
and since log-probabilities differ from logits / temperature by a per-step constant, the oracle's preference for b,
    margin = (log p_ref[b] - log q[b]) - (log p_ref[a] - log q[a])  >= 0,
can be at most 2 * eps_s / temperature, eps_s = max_v |logit_dev[s, v] - logit_ref[s, v]| (the MEASURED CFG-logit difference of
that step, itself bounded by the 2x-floor rule in the calling test).  A flip with a larger margin cannot come from logit
rounding: it is an indexing / schedule / sampler bug, and audit() fails on it.
"""
import torch


class RecordingMultinomial:
    """multinomial_fn for LLMOracle.sample: the draw of torch.multinomial(p.to(dev), 1, generator=gen), with p, q and the
    chosen id kept for every step."""

    def __init__(self, device, generator):
        self.dev, self.gen = device, generator
        self.p, self.q, self.ids = [], [], []

    def __call__(self, p):
        pd = p.to(self.dev)
        q = torch.empty_like(pd).exponential_(1.0, generator=self.gen)
        idx = torch.argmax(pd / q, dim=-1, keepdim=True)
        self.p.append(p.reshape(-1).double().cpu())
        self.q.append(q.reshape(-1).double().cpu())
        self.ids.append(int(idx))
        return idx.cpu()


def audit(dev_step_ids, rec: RecordingMultinomial, dev_logits, ref_logits, temperature=1.0, skip_steps=()):
    """dev_step_ids[s]: the id the device drew at decode step s (None where the schedule forced the token: nothing to compare);
    rec: the oracle's recorded draws (one per step); dev_logits / ref_logits [steps, V]: CFG logits of both sides.
    -> (n_compared, flips) with flips = [(step, dev_id, ref_id, margin, bound)], after asserting that every flip is explained."""
    assert len(rec.ids) == len(dev_step_ids) == dev_logits.shape[0] == ref_logits.shape[0], \
        (len(rec.ids), len(dev_step_ids), dev_logits.shape, ref_logits.shape)
    flips, n = [], 0
    for s, a in enumerate(dev_step_ids):
        if a is None or s in skip_steps:
            continue
        n += 1
        b = rec.ids[s]
        if a == b:
            continue
        p, q = rec.p[s], rec.q[s]
        assert p[a] > 0, f"step {s}: the device drew id {a}, which the oracle's distribution excludes (restriction / filter bug)"
        margin = float((torch.log(p[b]) - torch.log(q[b])) - (torch.log(p[a]) - torch.log(q[a])))
        eps = float((dev_logits[s].double() - ref_logits[s].double()).abs().max())
        bound = 2.0 * eps / temperature * (1 + 1e-3) + 1e-5
        assert -1e-9 <= margin <= bound, (f"step {s}: device id {a} vs oracle id {b}: the oracle prefers its id by {margin:.5f} in log p/q, "
                                          f"more than twice the step's measured logit difference allows ({bound:.5f}): not a rounding flip")
        flips.append((s, a, b, margin, bound))
    return n, flips
