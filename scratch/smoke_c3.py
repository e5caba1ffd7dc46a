

def main():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import __graft_entry__ as g
    from wu import _lib
    for v in (0, 2):
        _lib.call('wu_set_option', 5, v)
        print("c3 fwd variant", {0: "MFMA bf16", 2: "VALU fp32"}[v])
        g.smoke()


if __name__ == "__main__":
    main()
