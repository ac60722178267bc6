"""Empty stand-in: imported by the reference's download module, never called offline."""
