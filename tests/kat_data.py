"""Known-answer-test data of the reference's 1-D replicated illustration.

Regenerates the CASE 2 dataset of `illustration-examples/lcgp-rep-1d-illustration.ipynb`
(cells 5, 8, 12: skewed replication, seed 123) with the same numpy Generator call
sequence, and holds the numbers the notebook stores in its cell outputs (cells 12,
14, 18, 20).  These stored outputs are the only numeric pins the reference has.
"""
import numpy as np

# --- stored notebook outputs (cell 12 / 14 / 18 / 20) --------------------------------------
KAT_N_TOTAL = 194
KAT_N_UNIQUE = 40
KAT_FIRST_COUNTS = [1, 2, 1, 2, 2, 1, 2, 2]
KAT_DIAG_D = np.array([0.46745838, 0.75040147, 1.23211196])
KAT_VAR_G = np.array([0.96495696, 0.99189576, 0.9979089])
KAT_LENGTHSCALES = np.array([0.16848549, 0.26110824, 0.23634741])
KAT_LSIGMA2S = np.array([-6.11708833, -4.80458386, -4.6098362])
KAT_RMSE = 0.0233
KAT_NRMSE = 0.0343
KAT_COVER = 0.979
KAT_WIDTH = 0.1048
KAT_DSS = -20.3926


def truth(x):
    x = np.asarray(x, dtype=np.float64)
    return np.vstack([
        0.8 + 0.3 * np.sin(2 * np.pi * x) + 0.2 * x,
        0.3 + 0.5 * np.cos(2 * np.pi * x),
        -0.4 - (x - 0.5) ** 2 + 0.2 * np.sin(4 * np.pi * x),
    ])


def kat_dataset():
    """(xtrain (194,1), ytrain (3,194), xtest (400,1), ytrue (3,400))."""
    rng = np.random.default_rng(123)
    noise = (0.05, 0.08, 0.10)
    grid = np.linspace(0.0, 1.0, 40, dtype=np.float64)
    rows_x, rows_y = [], []
    for xi in grid:
        heavy = 0.20 <= xi <= 0.45
        count = int(rng.choice((8, 12, 16, 20) if heavy else (1, 2)))
        base = truth([xi])[:, 0]
        for _ in range(count):
            eps = np.array([rng.normal(0, s) for s in noise], dtype=np.float64)
            rows_x.append([xi])
            rows_y.append(base + eps)
    xtrain = np.array(rows_x, dtype=np.float64)
    ytrain = np.array(rows_y, dtype=np.float64).T
    xtest = np.linspace(0.0, 1.0, 400, dtype=np.float64)[:, None]
    return xtrain, ytrain, xtest, truth(xtest[:, 0])
