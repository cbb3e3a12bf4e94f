"""Workflow glue of the reference that sits directly on the TV operator (presets/workflows): the restoration
configuration contract and the factory that turns it into a ``darsia_amd.TVD``."""
