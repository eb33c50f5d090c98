"""Host logic of the EDM sampler that needs no GPU: which `denoiser` callables take the fused loop (nothing heuristic).
Reference closure: nsr/lsgm/sgm_DiffusionEngine.py:401-404 (`lambda input, sigma, c: self.denoiser(self.model, input, sigma, c, **kw)`)."""
from ln3diff_amd.sgm.sampling import DiscreteDenoiser, BoundDenoiser, _find_pair, _is_reference_lambda


class _Net:
    def prepare_context(self, ctx):
        return None

    def __call__(self, *a, **k):
        return None


class _Engine:
    def __init__(self):
        self.denoiser, self.model = DiscreteDenoiser(), _Net()


def test_reference_lambda_is_recognised_and_nothing_looser():
    self = _Engine()
    additional_model_inputs = {}
    ref = lambda input, sigma, c: self.denoiser(self.model, input, sigma, c, **additional_model_inputs)      # the reference's own text
    plain = lambda input, sigma, c: self.denoiser(self.model, input, sigma, c)
    assert _is_reference_lambda(ref) and _is_reference_lambda(plain)
    assert _find_pair(ref) == (self.denoiser, self.model) and _find_pair(plain) == (self.denoiser, self.model)
    # the same call with anything else around it runs as written (generic loop)
    clamp = lambda input, sigma, c: self.denoiser(self.model, input, sigma, c).clamp(-1, 1)
    scaled = lambda input, sigma, c: self.denoiser(self.model, input * 2, sigma, c)
    swapped = lambda input, sigma, c: self.denoiser(self.model, input, c, sigma)
    wrapped = lambda input, sigma, c: self.denoiser(lambda *a, **k: self.model(*a, **k), input, sigma, c)
    other_attr = lambda input, sigma, c: self.denoiser(self.model2, input, sigma, c)
    for f in (clamp, scaled, swapped, wrapped, other_attr):
        assert not _is_reference_lambda(f)
        assert _find_pair(f) == (None, None)
    extra = {'y': 1}
    with_kw = lambda input, sigma, c: self.denoiser(self.model, input, sigma, c, **extra)       # non-empty additional_model_inputs
    assert _find_pair(with_kw) == (None, None)

    def named(input, sigma, c):                           # a def with the same body is the same code
        return self.denoiser(self.model, input, sigma, c)
    assert _find_pair(named) == (self.denoiser, self.model)

    def two_statements(input, sigma, c):
        out = self.denoiser(self.model, input, sigma, c)
        return out
    assert _find_pair(two_statements) == (None, None)     # STORE_FAST: not the reference's expression


def test_explicit_routes():
    e = _Engine()
    assert _find_pair(e.denoiser.bind(e.model)) == (e.denoiser, e.model)
    assert _find_pair(e.denoiser.bind(e.model, y=1)) == (None, None)
    assert _find_pair(BoundDenoiser(e.denoiser, object())) == (None, None)            # not one of this package's networks

    class Opaque:
        def __call__(self, input, sigma, c):
            return e.denoiser(e.model, input, sigma, c)
    assert _find_pair(Opaque()) == (None, None)
    o = Opaque()
    o._ln3d_pair = (e.denoiser, e.model)                  # explicit opt-in
    assert _find_pair(o) == (e.denoiser, e.model)


def test_edm_discretization_is_the_karras_rho_schedule():
    """discretizer.py:27-39 on its published defaults: endpoints, monotone, the appended zero / flip plumbing shared with the legacy table."""
    import torch
    from ln3diff_amd.sgm.sampling import EDMDiscretization
    d = EDMDiscretization()
    s = d(10)
    assert s.shape == (11,) and float(s[-1]) == 0.0
    assert abs(float(s[0]) - 80.0) < 1e-4 and abs(float(s[9]) - 0.002) < 1e-7
    assert bool((s[:-1][1:] < s[:-1][:-1]).all())
    ramp = torch.linspace(0, 1, 10)
    ref = (80.0 ** (1 / 7.0) + ramp * (0.002 ** (1 / 7.0) - 80.0 ** (1 / 7.0))) ** 7.0
    assert torch.equal(d.get_sigmas(10), ref)
    assert torch.equal(d(10, do_append_zero=False, flip=True), torch.flip(ref, (0,)))
