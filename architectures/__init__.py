"""Drop-in package name of the reference (`architectures/`); implementation in cutmix-semisup-seg_amd/architectures/."""
