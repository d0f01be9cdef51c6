"""The backdoor attack's hook on the engine (reference backdoor.py:13-15, 52-65) -- SURVEY.md section 8(f).

The reference's `BackdoorAttack` subclasses `malicious.Attack` and overrides only `_attack_grads`: it trains a
malicious network starting from the parameters the honest mean step would reach, then asks for the gradient
that moves the server there, clipped to mean +- num_std * std so that it hides among the honest clients.
The training loop is the caller's (it is the reference's own torch code); the two vector steps around it run
in libbyzagg, bit-identical to the reference's numpy arithmetic:

    start     = original_params - lr * grads_mean                              backdoor.py:54
    mal       = self.train_malicious_network(start)                            backdoor.py:56   (caller)
    new_grads = clip(((start) - (mal + lr * grads_mean)) / lr, mean -+ z std)  backdoor.py:59-63

The reference's own `backdoor.py` also keeps working unchanged on top of the drop-in `malicious` module (its hook
receives host arrays); this class is for callers who want those two steps on the GPU, or device-resident.
"""
from . import malicious
from .engine import get_engine


class BackdoorAttack(malicious.Attack):
    def __init__(self, num_std, train_malicious_network=None):
        super(BackdoorAttack, self).__init__(num_std)
        if train_malicious_network is not None:
            self.train_malicious_network = train_malicious_network

    def train_malicious_network(self, initial_params_flat):
        raise NotImplementedError('supply the training loop of backdoor.py:83-135 (callable: start params -> '
                                  'trained params), as a constructor argument or by overriding this method')

    def _attack_grads(self, grads_mean, grads_stdev, original_params, learning_rate):
        eng = get_engine()
        initial_params_flat = eng.backdoor_initial_params(original_params, grads_mean, learning_rate)
        mal_net_params = self.train_malicious_network(initial_params_flat)
        return eng.backdoor_clip(grads_mean, grads_stdev, original_params, mal_net_params, learning_rate,
                                 self.num_std)
