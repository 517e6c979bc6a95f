"""fsv2v-mi355x: MI355X-native (gfx950) G/D training hot path of few-shot-vid2vid.

The directory name is the one the build contract asks for and is not a valid Python identifier; import it as

    import fsv2v_amd            # thin alias module at the repo root
"""
__version__ = "0.1.0"
