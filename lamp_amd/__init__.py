"""lamp_amd -- MI355X-native forward path of LaMP (label-graph message passing).

Mirrors the reference package layout (``lamp.Models``, ``lamp.Layers``, ``lamp.SubLayers`` ...);
``dropin/lamp`` re-exports these modules under the reference's package name.
"""
from . import Constants, utils, SubLayers, Attention, Layers, Encoders, Decoders, Models, Translator, Beam  # noqa: F401,E501
from .Models import LAMP  # noqa: F401
from . import data, evaluate, sharding  # noqa: F401,E402

__all__ = ['Constants', 'utils', 'SubLayers', 'Attention', 'Layers', 'Encoders', 'Decoders', 'Models',
           'Translator', 'Beam', 'LAMP']
