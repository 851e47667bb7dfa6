"""Drop-in alias: ``import fewbit`` resolves to the MI355X implementation in ``fewbit_amd`` with the module layout
of the reference package (``fewbit.functional``, ``fewbit.functional.activations.store``, ``fewbit.modules``,
``fewbit.util``)."""
import sys

import fewbit_amd
from fewbit_amd import *  # noqa: F401,F403
from fewbit_amd import functional, modules, util, approx, cli, compat, fft, linear, variance, store as _store_mod, map_module, memory_usage_hooks, __version__  # noqa: F401

sys.modules[__name__ + '.functional'] = functional
sys.modules[__name__ + '.functional.activations'] = functional
sys.modules[__name__ + '.modules'] = modules
sys.modules[__name__ + '.modules.activations'] = modules
sys.modules[__name__ + '.util'] = util
sys.modules[__name__ + '.approx'] = approx
sys.modules[__name__ + '.cli'] = cli
sys.modules[__name__ + '.compat'] = compat
sys.modules[__name__ + '.fft'] = fft
sys.modules[__name__ + '.functional.linear'] = linear
sys.modules[__name__ + '.modules.linear'] = linear
sys.modules[__name__ + '.functional.variance'] = variance
sys.modules[__name__ + '.modules.variance'] = variance
functional.linear = modules.linear = linear
functional.variance = modules.variance = variance
functional.activations = functional
modules.activations = modules
