def sec_render(): pass
def sec_decoder(): pass
