"""-m gpu: the legacy-layout shim -- the reference's sliding-window MMQR itself, run on the device for a given
window PR x PC, against the REAL reference's raw outputs: factored matrix (R above, reflector tails where qr.c:242-248 leaves them)
and the window-indexed tau array (qr.c:300-304).  Shape must be identical; values agree to rounding (the device sums in
wave-reduction order, the reference serially): 1e-12 of the matrix scale."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

GOLD = [("ref_6x4_f64_4x2", 6, 4, 4, 2), ("ref_128x32_f64_4x2", 128, 32, 4, 2), ("ref_120x32_f64_64x8", 120, 32, 64, 8),
        ("ref_64x20_f64_16x4", 64, 20, 16, 4)]


@pytest.mark.parametrize("name,m,n,PR,PC", GOLD)
def test_legacy_layout_vs_reference_golden(qr, oracle, name, m, n, PR, PC):
    """Fixtures hold the real reference's factored matrix, tau, Q and R for its own input (srand(12), qr.c:468-474)."""
    g = load_golden(name)
    A = oracle.fill_rand(m, n)
    assert np.array_equal(A, g["A"])
    F, tau = qr.mmqr_legacy(A, PR, PC)
    rp, cp = oracle.panel_dims(m, n, PR, PC)
    assert tau.shape == g["tau"].shape == (rp * cp * PC,)                     # qr.c:61 sizing, qr.c:300-304 indexing
    scale = np.abs(g["F"]).max()
    assert np.abs(F - g["F"]).max() <= 1e-12 * scale
    assert np.array_equal(tau == 0.0, g["tau"] == 0.0), "the same slots of the window-indexed array are filled"
    assert np.abs(tau - g["tau"]).max() <= 1e-12 * np.abs(g["tau"]).max()
    Q, R = qr.explicit_qr_legacy(F, tau, PR, PC)
    assert np.array_equal(R, np.triu(F))                                      # qr.c:334-343
    assert np.abs(Q - g["Q"]).max() <= 1e-12
    assert np.linalg.norm(A - Q @ R) / np.linalg.norm(A) < 1e-14 * np.sqrt(m) * 4
    assert np.linalg.norm(Q.T @ Q - np.eye(m)) < 10 * max(float(g["orth"]), 1e-14)


@pytest.mark.parametrize("m,n,PR,PC", [(512, 128, 64, 8), (512, 128, 4, 2), (508, 128, 16, 4), (232, 64, 64, 8), (1016, 512, 64, 8),
                                       (304, 48, 64, 16), (64, 64, 64, 8)])
def test_legacy_layout_vs_oracle(qr, oracle, m, n, PR, PC):
    """BASELINE config C1 (512 x 128) and other window shapes against the oracle restatement (bitwise-pinned to the real reference):
    factored matrix and tau element by element, then the reference's own self-check flow (explicitQR, Q R = A)."""
    oracle.check_shape(m, n, PR, PC)
    A = oracle.fill_rand(m, n)
    Fo, tauo, _ = oracle.mmqr(A, PR, PC)
    F, tau = qr.mmqr_legacy(A, PR, PC)
    assert tau.shape == tauo.shape
    assert np.abs(F - Fo).max() <= 1e-12 * np.abs(Fo).max()
    assert np.abs(tau - tauo).max() <= 1e-12 * np.abs(tauo).max()
    if m <= 512:
        Q, R = qr.explicit_qr_legacy(F, tau, PR, PC)
        assert np.linalg.norm(A - Q @ R) / np.linalg.norm(A) < 1e-13
        assert np.linalg.norm(Q.T @ Q - np.eye(m)) < 1e-11
        # and the blocked path's R is the same R (sign-normalised): the two entry points factor the same matrix
        F2, _ = qr.mmqr(A)
        assert np.linalg.norm(oracle.sign_normalise(F) - oracle.sign_normalise(F2)) / np.linalg.norm(np.triu(F2[:n])) < 1e-13


def test_legacy_layout_rejects_what_the_reference_cannot_factor(qr):
    A = np.random.default_rng(0).random((100, 32))
    for PR, PC in ((64, 8), (128, 8), (64, 3), (8, 8)):         # (100 - 64) % 56 != 0; PR > 64; PC not a supported width; PC = PR
        with pytest.raises(qr.QRError, match="invalid argument"):
            qr.mmqr_legacy(A, PR, PC)
