"""MI355X-native early-exit Conformer encoder (drop-in for the reference's
``models.model.early_exit.Early_conformer`` / ``full_conformer`` encoder path; ``model.Splitformer`` and
``model.Early_zipformer`` are composed from the same kernels)."""
__version__ = "0.1.0"
