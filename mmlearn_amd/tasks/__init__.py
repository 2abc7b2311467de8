"""Task (LightningModule) surface of the hot path."""
