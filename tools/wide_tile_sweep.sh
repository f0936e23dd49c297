CMS_VARIANTS=0:0,0:51,256:61,256:60,256:62,0:71,0:72,0:73,0:70 python tools/conv_variants.py "c2 l3" "c2 l4" "c3 l3" "c3 l4"
