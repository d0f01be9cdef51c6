"""`import defences` shim: put this directory first on sys.path and the reference's main.py / server.py run
unchanged against the MI355X engine."""
from attacking_federate_learning_amd.defences import *  # noqa: F401,F403
from attacking_federate_learning_amd.defences import _krum_create_distances, defend, DefenseTypes  # noqa: F401
