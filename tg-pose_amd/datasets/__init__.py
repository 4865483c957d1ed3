"""Drop-in pieces of the reference's ``datasets`` package that run on the device."""
