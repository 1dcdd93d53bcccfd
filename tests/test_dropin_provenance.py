"""CPU: which renders join the batch of an iteration in the C++ half of the drop-in package (Batch::matches, dgr_native.cpp).  The
reference's scene model computes exp / sigmoid / normalize / cat anew on every getter call
(/root/reference/thirdparty/gaussian_splatting/scene/gaussian_model.py:76-101) and render() calls the getters per view
(gaussian_renderer/__init__.py:89-111): the twelve renders of an iteration hand in different tensor objects with identical values.
They must batch; anything whose values COULD differ must not."""
import pytest
import torch

F = torch.nn.functional


@pytest.fixture(scope="module")
def ext():
    import diff_gaussian_rasterization as drg
    e = drg.native_extension()
    if e is None:
        pytest.skip("diff_gaussian_rasterization/_dgr.so is not built")
    return e


def _leaves():
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g).requires_grad_()
    return dict(rot=r(7, 4), scale=r(7, 1), opac=r(7, 1), f_dc=r(7, 1, 3), f_rest=torch.zeros(7, 0, 3).requires_grad_())


def _getters(p):
    """The reference's getters + render()'s isotropic-scale repeat (gaussian_model.py:76-101, gaussian_renderer/__init__.py:89-93)."""
    return dict(rot=F.normalize(p["rot"]), scale=torch.exp(p["scale"]).repeat(1, 3), opac=torch.sigmoid(p["opac"]),
                sh=torch.cat((p["f_dc"], p["f_rest"]), dim=1))


def test_reference_getters_of_two_renders_share_a_batch(ext):
    p = _leaves()
    first, second = _getters(p), _getters(p)
    for k in first:
        assert first[k] is not second[k] and torch.equal(first[k], second[k])
        ext.provenance_open(first[k])
        assert ext.provenance_joins(second[k]), k
        assert ext.provenance_joins(first[k]), k                 # (identity still matches)


def test_a_leaf_is_only_matched_by_itself(ext):
    p = _leaves()
    ext.provenance_open(p["rot"])
    assert ext.provenance_joins(p["rot"])
    assert not ext.provenance_joins(p["rot"].detach().clone().requires_grad_())


def test_what_could_hold_other_values_does_not_batch(ext):
    p = _leaves()
    ext.provenance_open(torch.sigmoid(p["opac"]))
    assert not ext.provenance_joins(torch.exp(p["opac"]))                          # another operation
    assert not ext.provenance_joins(torch.sigmoid(p["scale"]))                     # another leaf
    assert not ext.provenance_joins(torch.sigmoid(p["opac"]).detach())             # no graph at all
    ext.provenance_open(torch.exp(p["scale"]) * 2.0)                               # an operation with a hidden scalar: never by provenance
    assert not ext.provenance_joins(torch.exp(p["scale"]) * 2.0)
    ext.provenance_open(torch.exp(p["scale"]) * torch.ones(7, 1))                  # an operand without a graph
    assert not ext.provenance_joins(torch.exp(p["scale"]) * torch.ones(7, 1))
    ext.provenance_open(torch.exp(p["scale"]).repeat(1, 3))
    assert not ext.provenance_joins(torch.exp(p["scale"]))                         # another shape / chain


def test_a_parameter_update_between_two_renders_ends_the_batch(ext):
    p = _leaves()
    ext.provenance_open(torch.exp(p["scale"]))
    assert ext.provenance_joins(torch.exp(p["scale"]))
    with torch.no_grad():
        p["scale"].add_(0.1)                                                       # (an optimiser step: bumps the leaf's version)
    assert not ext.provenance_joins(torch.exp(p["scale"]))


def test_chains_that_differ_only_in_a_non_tensor_argument_do_not_batch(ext):
    """node names + topology + leaf versions are not enough -- the whitelisted operators also depend on dims, sizes, a
    clamp minimum, a norm order.  Same names, same leaves, same final shape, other values: must not join."""
    g = torch.Generator().manual_seed(2)
    sq = torch.randn(4, 4, generator=g).requires_grad_()
    ext.provenance_open(torch.exp(sq.permute(1, 0)).contiguous())
    assert ext.provenance_joins(torch.exp(sq.permute(1, 0)).contiguous())
    assert not ext.provenance_joins(torch.exp(sq.permute(0, 1)).contiguous())                # another permutation of a square tensor
    p = _leaves()
    ext.provenance_open(F.normalize(p["rot"]))
    assert ext.provenance_joins(F.normalize(p["rot"]))
    assert not ext.provenance_joins(F.normalize(p["rot"], eps=1e-3))                         # another clamp minimum
    assert not ext.provenance_joins(F.normalize(p["rot"], p=1.0))                            # another norm order
    r4 = torch.randn(4, 4, generator=g).requires_grad_()
    ext.provenance_open(F.normalize(r4, dim=1))
    assert not ext.provenance_joins(F.normalize(r4, dim=0))                                  # another dim, same shapes throughout
    c1, c2 = torch.randn(3, 3, generator=g).requires_grad_(), torch.randn(3, 3, generator=g).requires_grad_()
    ext.provenance_open(torch.cat((c1, c2), dim=0).view(3, 6))
    assert ext.provenance_joins(torch.cat((c1, c2), dim=0).view(3, 6))
    assert not ext.provenance_joins(torch.cat((c1, c2), dim=1).view(3, 6))                   # cat along another dim, viewed to the same shape
    ext.provenance_open(torch.exp(p["scale"]).repeat(1, 3).view(3, 7))
    assert not ext.provenance_joins(torch.exp(p["scale"]).repeat(3, 1).view(3, 7))           # other repeats, same final shape
