"""Drop-in shim: makes ``import lamp.Models`` / ``from lamp.Models import LAMP`` (what the reference's
main.py, test.py and train.py do) resolve to the MI355X-native implementation in ``lamp_amd``.

Put this directory's parent (``<repo>/dropin``) and the repo root on PYTHONPATH *instead of* the
reference's own ``lamp`` package; see INTEGRATION.md.
"""
import sys

import lamp_amd
from lamp_amd import (Attention, Beam, Constants, Decoders, Encoders, Layers, Models, SubLayers,  # noqa: F401
                      Translator, utils)

for _name in ('Attention', 'Beam', 'Constants', 'Decoders', 'Encoders', 'Layers', 'Models', 'SubLayers',
              'Translator', 'utils'):
    sys.modules[__name__ + '.' + _name] = getattr(lamp_amd, _name)

__all__ = [Constants, Layers, SubLayers, Models, Translator, Beam, Encoders, Decoders]
