"""Drop-in replacement for the reference's `defences` module (reference defences.py:1-75).

Same names, same call signatures, same return conventions; the arithmetic runs in libbyzagg's HIP kernels
on an MI355X.  `Server.defend` (reference server.py:87) calls

    defences.defend[defence_method](self.users_grads, len(self.users), int(len(self.users)*self.mal_prop))

with a C-contiguous np.float32 matrix and gets a 1-D np.float32 vector back.  Device-resident inputs
(torch CUDA tensors) are accepted too and then nothing crosses PCIe.

Differences a caller can observe, all documented in DESIGN.md:
  * `krum` on a host numpy matrix returns a VIEW of the winning row, as the reference does (defences.py:42); on a
    device-resident matrix it returns a copy made on the device, so that the index need not cross to the host;
  * `_krum_create_distances` returns a `Distances` handle (GPU-resident N x N matrix) instead of a dict
    of dicts; `krum(..., distances=handle)` accepts it, `handle.to_dict()` rebuilds the reference's form.
"""
from .engine import Distances, get_engine  # noqa: F401


class DefenseTypes:
    NoDefense = 'NoDefense'
    Krum = 'Krum'
    TrimmedMean = 'TrimmedMean'
    Bulyan = 'Bulyan'

    def __str__(self):
        return self.value


def no_defense(users_grads, users_count, corrupted_count):
    """Column mean of the gradient matrix (reference defences.py:13-14)."""
    return get_engine().no_defense(users_grads, users_count, corrupted_count)


def _krum_create_distances(users_grads):
    """All pairwise client distances (reference defences.py:16-21), kept on the GPU."""
    return get_engine().pairwise_distances(users_grads)


def krum(users_grads, users_count, corrupted_count, distances=None, return_index=False, debug=False):
    """Krum as the reference defines it (defences.py:23-42): unsquared norms, the n-f smallest summed,
    candidates visited in the order 1, 0, 2, ... with a strict '<'."""
    return get_engine().krum(users_grads, users_count, corrupted_count, distances=distances,
                             return_index=return_index)


def trimmed_mean(users_grads, users_count, corrupted_count):
    """Mean of the k = rows - corrupted - 1 values closest to the median, per parameter
    (reference defences.py:44-52)."""
    return get_engine().trimmed_mean(users_grads, users_count, corrupted_count)


def bulyan(users_grads, users_count, corrupted_count):
    """Bulyan (reference defences.py:55-70): n - 2f iterated Krum picks, then the trimmed mean of the picked
    rows in selection order."""
    return get_engine().bulyan(users_grads, users_count, corrupted_count)


defend = {DefenseTypes.Krum: krum,
          DefenseTypes.TrimmedMean: trimmed_mean, DefenseTypes.NoDefense: no_defense,
          DefenseTypes.Bulyan: bulyan}
