from .adm import AdmUnet2d  # noqa: F401
