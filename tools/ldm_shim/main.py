from ldm.util import instantiate_from_config  # re-export needed by rdm/modules/diffusionmodules/openaimodel.py:596
