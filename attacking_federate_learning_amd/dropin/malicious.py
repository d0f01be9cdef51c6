"""`import malicious` shim: see dropin/defences.py."""
from attacking_federate_learning_amd.malicious import Attack, DriftAttack  # noqa: F401
