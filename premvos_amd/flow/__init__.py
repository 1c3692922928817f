from .pwcnet import PWCDCNet, pwc_dc_net  # noqa: F401
