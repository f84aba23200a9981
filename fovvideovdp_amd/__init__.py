"""MI355X-native FovVideoVDP hot path."""
