"""`custom_imports = dict(imports=['attentionshift_amd.mmdet_plugin'])` in a reference config registers the MI355X
classes under the reference's names in mmdet's own BACKBONES / HEADS registries.
Importing this module without mmdet installed raises ImportError, as a custom import should."""
from . import register_into_mmdet

register_into_mmdet()
