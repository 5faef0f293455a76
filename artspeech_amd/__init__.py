"""MI355X-native ArtSpeech acoustic-model inference path."""
__version__ = "0.1.0"
