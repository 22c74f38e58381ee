"""`specdec` command line (counterpart of the reference's src/specdec_cli)."""
