"""Spectral helpers of the host side.

`convert_spectrum` re-bins a tabulated spectrum onto the model's wavelength bins by integrating the piecewise
interpolant over every bin (reference source/tools.py:116-288, used for cloud cross-sections
(source/clouds.py:118-120) and stellar spectra).  `type="linear"`: trapezoids of the linear interpolant;
`type="log"`: the same in log(flux), i.e. geometric means weighted by wavelength distance.  Bins that are not
fully inside the tabulated range get 0.
"""
import numpy as np


def _interface_values(old_lambda, old_flux, int_lambda, log):
    """interpolant at the bin interfaces; 0 marks "outside the tabulated range" (tools.py:177-193, :236-252)"""
    out = np.zeros(len(int_lambda))
    for i, lam in enumerate(int_lambda):
        if lam < old_lambda[0]:
            continue
        if lam > old_lambda[-1]:
            break
        p = int(np.searchsorted(old_lambda, lam, side="left")) - 1      # last tabulated point below lam
        d_hi, d_lo, width = old_lambda[p + 1] - lam, lam - old_lambda[p], old_lambda[p + 1] - old_lambda[p]
        if log:
            out[i] = (old_flux[p] ** d_hi * old_flux[p + 1] ** d_lo) ** (1 / width)
        else:
            out[i] = (old_flux[p] * d_hi + old_flux[p + 1] * d_lo) / width
    return out


def convert_spectrum(old_lambda, old_flux, new_lambda, int_lambda=None, type="linear", extrapolate_with_BB_T=0):
    if extrapolate_with_BB_T != 0:
        raise NotImplementedError("black-body extrapolation of re-binned spectra is a star-tool feature (SURVEY.md 2.1)")
    if type not in ("linear", "log"):
        raise ValueError("type must be 'linear' or 'log'")
    old_lambda = np.asarray(old_lambda, float)
    old_flux = np.asarray(old_flux, float)
    new_lambda = np.asarray(new_lambda, float)
    if int_lambda is None:
        mid = 0.5 * (new_lambda[1:] + new_lambda[:-1])
        int_lambda = np.concatenate(([new_lambda[0] - (new_lambda[1] - new_lambda[0]) / 2], mid,
                                     [new_lambda[-1] + (new_lambda[-1] - new_lambda[-2]) / 2]))
    int_lambda = np.asarray(int_lambda, float)
    log = type == "log"
    edge = _interface_values(old_lambda, old_flux, int_lambda, log)
    new_flux = []
    for i in range(len(new_lambda)):
        lo, hi = int_lambda[i], int_lambda[i + 1]
        if edge[i] == 0 or edge[i + 1] == 0:
            new_flux.append(0.0)
            continue
        first = int(np.searchsorted(old_lambda, lo, side="left"))      # first tabulated point >= lo
        if not old_lambda[first] < hi:
            # no tabulated point inside the bin: mean of the two interface values
            new_flux.append((edge[i] * edge[i + 1]) ** 0.5 if log else (edge[i] + edge[i + 1]) / 2.0)
            continue
        # nodes of the integrand inside the bin: lo, the tabulated points in [lo, hi), hi
        last = int(np.searchsorted(old_lambda, hi, side="left"))        # first tabulated point >= hi
        if last >= len(old_lambda):
            # the table ends inside the bin: the reference leaves its running value unnormalised here; this can
            # only happen when hi coincides with the last tabulated wavelength to rounding -- treat as outside
            new_flux.append(0.0)
            continue
        x = np.concatenate(([lo], old_lambda[first:last], [hi]))
        y = np.concatenate(([edge[i]], old_flux[first:last], [edge[i + 1]]))
        acc = 1.0 if log else 0.0
        for k in range(len(x) - 1):
            if log:
                acc *= (y[k] * y[k + 1]) ** (0.5 * (x[k + 1] - x[k]))
            else:
                acc += (y[k] + y[k + 1]) / 2.0 * (x[k + 1] - x[k])
        new_flux.append(acc ** (1 / (hi - lo)) if log else acc / (hi - lo))
    return new_flux
